"""ComfyUI V3 node classes: `Video Stabilizer Flow` and `Video Stabilizer Motion Apply`.

Socket ids, order, defaults and output names are those of the reference
(nodes/video_stabilizer_flow.py:643-763, nodes/video_stabilizer_motion_apply.py:29-129; pinned by
scripts/check_node_schema.py:28-64) so existing graphs keep working when this package replaces it.
"""

from __future__ import annotations

from typing import Any

from . import host_math as hm
from .apply_pipeline import apply_motion
from .comfy_compat import ComfyExtension, ProgressBar, io
from .flow_pipeline import _stabilize_frames

JSONType = io.Custom("JSON")

BLUR_QUALITY_SAMPLES = {"Draft": 5, "Standard": 9, "High": 17, "Ultra": 33}  # motion_apply node :21-26


class VideoStabilizerFlow(io.ComfyNode):
    """Dense-flow (DIS) stabilizer; all pixel work runs on the MI355X through libvstab."""

    @classmethod
    def define_schema(cls) -> io.Schema:
        schema = io.Schema(
            node_id="video_stabilizer_flow",
            display_name="Video Stabilizer Flow",
            category="Video/Stabilization",
            description=(
                "Video stabilization using dense optical flow with configurable transforms and framing, "
                "emitting stabilized frames, a padding mask, and motion diagnostics (MI355X build)."
            ),
        )
        slider = io.NumberDisplay.slider
        schema.inputs = [
            io.Image.Input("frames", display_name="Frames"),
            io.Float.Input("frame_rate", default=16.0, min=1.0, step=0.1, display_name="Input FPS",
                           tooltip="Frame rate in frames per second used to scale smoothing window."),
            io.Combo.Input("framing_mode", options=["crop", "crop_and_pad", "expand"], default="crop_and_pad",
                           display_name="Framing Mode",
                           tooltip="Choose how borders produced by stabilization are handled."),
            io.Combo.Input("transform_mode", options=["translation", "similarity", "perspective"],
                           default="similarity", display_name="Transform Mode",
                           tooltip="Select the geometric model fitted to the optical flow."),
            io.Boolean.Input("camera_lock", default=False, display_name="Camera Lock",
                             tooltip="Aggressively pull the motion curve toward a locked tripod-like solution."),
            io.Float.Input("strength", default=0.7, min=0.0, max=1.0, step=0.05, display_name="Strength",
                           tooltip="Removal gain (0 keeps original motion, 1 removes it using the smoothed motion curve).",
                           display_mode=slider),
            io.Float.Input("smooth", default=0.5, min=0.0, max=1.0, step=0.05, display_name="Smooth",
                           tooltip="Temporal smoothing amount applied to the motion curve before removal.",
                           display_mode=slider),
            io.Float.Input("keep_fov", default=0.6, min=0.0, max=1.0, step=0.05, display_name="Keep FOV",
                           tooltip=("[Crop only] How much of the original FOV to preserve (1.0 = no zoom, 0.0 = maximum "
                                    "zoom). Ignored when framing_mode is crop_and_pad or expand."),
                           display_mode=slider),
            io.Color.Input("padding_color", default="#7F7F7F", display_name="Padding Color",
                           tooltip="HEX padding color applied in crop_and_pad / expand (e.g. #404040)."),
        ]
        schema.outputs = [
            io.Image.Output("frames_stabilized", display_name="Stabilized Frames"),
            io.Mask.Output("padding_mask", display_name="Padding Mask"),
            JSONType.Output("meta", display_name="Motion Meta"),
        ]
        return schema

    @classmethod
    def execute(cls, frames: Any, frame_rate: float, framing_mode: str, transform_mode: str, camera_lock: bool,
                strength: float, smooth: float, keep_fov: float, padding_color: str) -> io.NodeOutput:
        context = hm._normalize_video_input(frames)
        result = _stabilize_frames(
            context, framing_mode, transform_mode, camera_lock, strength, smooth, keep_fov,
            hm._parse_padding_color(padding_color), frame_rate, keep_on_device=True,
        )
        return io.NodeOutput(hm._reconstruct_video(result.frames, context),
                             hm._convert_masks_for_output(result.masks), result.meta)


class VideoStabilizerMotionApply(io.ComfyNode):
    """Apply motion_meta matrices (optionally with matrix-sampled motion blur) to a clip."""

    @classmethod
    def define_schema(cls) -> io.Schema:
        schema = io.Schema(
            node_id="video_stabilizer_motion_apply",
            display_name="Video Stabilizer Motion Apply",
            category="Video/Stabilization",
            description="Applies motion metadata to frames and emits a padding mask.",
        )
        schema.inputs = [
            io.Image.Input("frames", display_name="Frames"),
            JSONType.Input("motion_meta", display_name="Motion Meta"),
            io.Combo.Input("framing_mode", options=["crop_and_pad", "crop", "expand"], default="crop_and_pad",
                           display_name="Framing Mode"),
            io.Combo.Input("interpolation", options=["bilinear", "bicubic"], default="bilinear",
                           display_name="Interpolation"),
            io.Color.Input("padding_color", default="#7F7F7F", display_name="Padding Color",
                           tooltip="HEX padding color used where warping exposes empty pixels."),
            io.Float.Input("motion_blur", default=0.0, min=0.0, max=1.0, step=0.05, display_name="Motion Blur",
                           tooltip="Shutter fraction for matrix-sampled motion blur. 0 disables blur.",
                           display_mode=io.NumberDisplay.slider),
            io.Combo.Input("motion_blur_quality", options=list(BLUR_QUALITY_SAMPLES.keys()), default="Standard",
                           display_name="Blur Quality",
                           tooltip="Draft is faster. High and Ultra average more shutter samples for smoother blur."),
        ]
        schema.outputs = [
            io.Image.Output("frames", display_name="Frames"),
            io.Mask.Output("padding_mask", display_name="Padding Mask"),
            JSONType.Output("meta", display_name="Meta"),
        ]
        return schema

    @classmethod
    def execute(cls, frames: Any, motion_meta: dict, framing_mode: str, interpolation: str, padding_color: str,
                motion_blur: float, motion_blur_quality: str) -> io.NodeOutput:
        context = hm._normalize_video_input(frames)
        quality = motion_blur_quality if motion_blur_quality in BLUR_QUALITY_SAMPLES else "Standard"
        samples = BLUR_QUALITY_SAMPLES[quality]
        n = len(context.frames)
        per_frame = int(max(3, min(33, samples))) if motion_blur > 0.0 else 1
        total = max(n * per_frame + (n if framing_mode == "crop" else 0), 1)
        pbar = ProgressBar(total)
        done = 0

        def tick() -> None:
            nonlocal done
            done += 1
            pbar.update_absolute(min(done, total), total)

        result = apply_motion(context, motion_meta, hm._parse_padding_color(padding_color),
                              framing_mode=framing_mode, interpolation=interpolation, motion_blur=motion_blur,
                              motion_blur_samples=samples, progress_callback=tick, keep_on_device=True)
        result.meta.setdefault("motion_apply", {})["motion_blur_quality"] = quality
        pbar.update_absolute(total, total)
        return io.NodeOutput(hm._reconstruct_video(result.frames, context),
                             hm._convert_masks_for_output(result.masks), result.meta)


NODE_CLASSES = [VideoStabilizerFlow, VideoStabilizerMotionApply]


class VideoStabilizerAmdExtension(ComfyExtension):
    async def get_node_list(self) -> list:
        return list(NODE_CLASSES)
