"""ComfyUI V3 node classes: `Video Stabilizer Flow` and `Video Stabilizer Motion Apply`.

Socket ids, order, defaults and output names are those of the reference
(nodes/video_stabilizer_flow.py:643-763, nodes/video_stabilizer_motion_apply.py:29-129; pinned by
scripts/check_node_schema.py:28-64) so existing graphs keep working when this package replaces it.
"""

from __future__ import annotations

from typing import Any

import os

from . import host_math as hm
from .apply_pipeline import apply_motion
from .meta_v2 import resolve_motion_meta
from .comfy_compat import ComfyExtension, ProgressBar, io
from .flow_pipeline import _stabilize_frames
from .shake_generator import STYLES, ShakeRecipe, generate_shake_motion_meta

JSONType = io.Custom("JSON")

BLUR_QUALITY_SAMPLES = {"Draft": 5, "Standard": 9, "High": 17, "Ultra": 33}  # motion_apply node :21-26


def _keep_on_device() -> bool:
    """with VSTAB_KEEP_ON_DEVICE=1 the IMAGE / MASK outputs stay in HBM (torch device tensors), so a
    Flow -> Motion Apply chain avoids the 15 GB host round trip of a 256x1080p clip.  Default: CPU tensors, as
    the reference returns (stabilizer_utils.py:200-221)."""
    return os.environ.get("VSTAB_KEEP_ON_DEVICE", "0") not in ("", "0", "false", "False")


def _image_out(frames, context):
    if _keep_on_device() and hasattr(frames, "device") and context.template_kind != "dict":
        return frames
    return hm._reconstruct_video(frames, context)


def _mask_out(masks, levels: int = 1):
    """levels: the motion-blur samples S behind a soft mask (its values are 1 - c / S: they may cross PCIe as bytes), 1 otherwise."""
    if _keep_on_device() and hasattr(masks, "device"):
        return masks[..., 0] if masks.ndim == 4 else masks
    return hm._convert_masks_for_output(masks, levels)


def _estimator_inputs(transform_tip: str, lock_tip: str, framing_tip: str, strength_tip: str, smooth_tip: str) -> list:
    """The nine input sockets shared by the Flow and Classic nodes (flow.py:657-726, classic.py:586-659):
    same ids, order, defaults and ranges; only some tooltips differ."""
    slider = io.NumberDisplay.slider
    return [
        io.Image.Input("frames", display_name="Frames"),
        io.Float.Input("frame_rate", default=16.0, min=1.0, step=0.1, display_name="Input FPS",
                       tooltip="Frame rate in frames per second used to scale smoothing window."),
        io.Combo.Input("framing_mode", options=["crop", "crop_and_pad", "expand"], default="crop_and_pad",
                       display_name="Framing Mode", tooltip=framing_tip),
        io.Combo.Input("transform_mode", options=["translation", "similarity", "perspective"],
                       default="similarity", display_name="Transform Mode", tooltip=transform_tip),
        io.Boolean.Input("camera_lock", default=False, display_name="Camera Lock", tooltip=lock_tip),
        io.Float.Input("strength", default=0.7, min=0.0, max=1.0, step=0.05, display_name="Strength",
                       tooltip=strength_tip, display_mode=slider),
        io.Float.Input("smooth", default=0.5, min=0.0, max=1.0, step=0.05, display_name="Smooth",
                       tooltip=smooth_tip, display_mode=slider),
        io.Float.Input("keep_fov", default=0.6, min=0.0, max=1.0, step=0.05, display_name="Keep FOV",
                       tooltip=("[Crop only] How much of the original FOV to preserve (1.0 = no zoom, 0.0 = maximum "
                                "zoom). Ignored when framing_mode is crop_and_pad or expand."),
                       display_mode=slider),
        io.Color.Input("padding_color", default="#7F7F7F", display_name="Padding Color",
                       tooltip="HEX padding color applied in crop_and_pad / expand (e.g. #404040)."),
    ]


def _estimator_outputs() -> list:
    return [
        io.Image.Output("frames_stabilized", display_name="Stabilized Frames"),
        io.Mask.Output("padding_mask", display_name="Padding Mask"),
        JSONType.Output("meta", display_name="Motion Meta"),
    ]


def _run_estimator_node(estimator: str, frames: Any, frame_rate: float, framing_mode: str, transform_mode: str,
                        camera_lock: bool, strength: float, smooth: float, keep_fov: float, padding_color: str):
    context = hm._normalize_video_input(frames)
    result = _stabilize_frames(
        context, framing_mode, transform_mode, camera_lock, strength, smooth, keep_fov,
        hm._parse_padding_color(padding_color), frame_rate, keep_on_device=True, estimator=estimator,
    )
    return io.NodeOutput(_image_out(result.frames, context), _mask_out(result.masks), result.meta)


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
        schema.inputs = _estimator_inputs(
            transform_tip="Select the geometric model fitted to the optical flow.",
            lock_tip="Aggressively pull the motion curve toward a locked tripod-like solution.",
            framing_tip="Choose how borders produced by stabilization are handled.",
            strength_tip="Removal gain (0 keeps original motion, 1 removes it using the smoothed motion curve).",
            smooth_tip="Temporal smoothing amount applied to the motion curve before removal.",
        )
        schema.outputs = _estimator_outputs()
        return schema

    @classmethod
    def execute(cls, frames: Any, frame_rate: float, framing_mode: str, transform_mode: str, camera_lock: bool,
                strength: float, smooth: float, keep_fov: float, padding_color: str) -> io.NodeOutput:
        return _run_estimator_node("flow", frames, frame_rate, framing_mode, transform_mode, camera_lock, strength,
                                   smooth, keep_fov, padding_color)


class VideoStabilizerClassic(io.ComfyNode):
    """Sparse feature-tracking stabilizer (corner detection + pyramidal LK, classic.py:69-160) -- same sockets
    and meta as the reference's `Video Stabilizer Classic`; the tracker runs on the MI355X as well."""

    @classmethod
    def define_schema(cls) -> io.Schema:
        schema = io.Schema(
            node_id="video_stabilizer_classic",
            display_name="Video Stabilizer Classic",
            category="Video/Stabilization",
            description=(
                "Video stabilization using sparse feature tracking with configurable transforms and framing, "
                "emitting both stabilized frames and a padding mask (MI355X build)."
            ),
        )
        schema.inputs = _estimator_inputs(
            transform_tip="Select the geometric model used to estimate camera motion.",
            lock_tip="Treat the shot as tripod-like by aggressively damping motion.",
            framing_tip="Choose how to handle borders produced by stabilization.",
            strength_tip="Removal gain (0 keeps original motion, 1 removes it based on smoothing).",
            smooth_tip="Temporal smoothing amount applied to the estimated motion path.",
        )
        schema.outputs = _estimator_outputs()
        return schema

    @classmethod
    def execute(cls, frames: Any, frame_rate: float, framing_mode: str, transform_mode: str, camera_lock: bool,
                strength: float, smooth: float, keep_fov: float, padding_color: str) -> io.NodeOutput:
        return _run_estimator_node("classic", frames, frame_rate, framing_mode, transform_mode, camera_lock, strength,
                                   smooth, keep_fov, padding_color)


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
        ma = result.meta.get("motion_apply", {})
        levels = int(ma.get("motion_blur_samples", 1)) if float(ma.get("motion_blur", 0.0)) > 0.0 else 1
        return io.NodeOutput(_image_out(result.frames, context), _mask_out(result.masks, levels), result.meta)


class VideoStabilizerInverse(io.ComfyNode):
    """Deprecated thin wrapper kept for graph compatibility (nodes/video_stabilizer_inverse.py:26-93 of the
    reference): inverse of the recorded stabilization warp through Motion Apply (crop_and_pad, bilinear)."""

    @classmethod
    def define_schema(cls) -> io.Schema:
        schema = io.Schema(
            node_id="video_stabilizer_inverse",
            display_name="Video Stabilizer Inverse",
            category="Video/Stabilization",
            description=("Deprecated: use Video Stabilizer Motion Apply. Restores stabilized frames to the original "
                         "canvas using stabilization metadata, and emits a padding mask for areas without source pixels."),
            is_deprecated=True,
        )
        schema.inputs = [
            io.Image.Input("frames", display_name="Frames"),
            JSONType.Input("meta", display_name="Meta"),
            io.Color.Input("padding_color", default="#7F7F7F", display_name="Padding Color",
                           tooltip="HEX padding color used where inverse warping exposes empty pixels."),
        ]
        schema.outputs = [
            io.Image.Output("frames_restored", display_name="Restored Frames"),
            io.Mask.Output("padding_mask", display_name="Padding Mask"),
            JSONType.Output("meta", display_name="Meta"),
        ]
        return schema

    @classmethod
    def execute(cls, frames: Any, meta: dict, padding_color: str) -> io.NodeOutput:
        context = hm._normalize_video_input(frames)
        legacy = dict(meta)
        legacy.pop("motion_meta", None)   # force the inverse of stabilization_warp
        motion = resolve_motion_meta(legacy)
        result = apply_motion(context, legacy, hm._parse_padding_color(padding_color), framing_mode="crop_and_pad",
                              interpolation="bilinear", keep_on_device=True)
        if isinstance(meta, dict) and isinstance(meta.get("motion_meta"), dict):
            result.meta["motion_meta"] = meta["motion_meta"]
        result.meta.pop("motion_apply", None)
        warp = meta.get("stabilization_warp", {}) if isinstance(meta, dict) else {}
        result.meta["inverse_stabilization"] = {
            "source_size": [int(motion.output_size[0]), int(motion.output_size[1])],
            "input_size": [int(motion.input_size[0]), int(motion.input_size[1])],
            "output_size": [int(motion.output_size[0]), int(motion.output_size[1])],
            "matrix_convention": "stabilized_to_source",
            "source_matrix_convention": "source_to_stabilized",
            "framing_mode": warp.get("framing_mode") if isinstance(warp, dict) else None,
            "note": "Restores original motion/canvas; pixels discarded by crop framing cannot be recovered.",
        }
        return io.NodeOutput(_image_out(result.frames, context), _mask_out(result.masks), result.meta)


def _shake_common_inputs(middle: list) -> list:
    """`frames_context`, `frame_rate`, <node-specific sockets>, `amount`, `speed`, `seed`
    (video_stabilizer_shake_generator.py:27-79, video_stabilizer_shake_generator_manual.py:29-136)."""
    slider = io.NumberDisplay.slider
    return [
        io.Image.Input("frames_context", display_name="Frames Context",
                       tooltip=("The input frames are used only to read frame count and resolution. This node outputs "
                                "motion metadata only; connect it to Video Stabilizer Motion Apply to move pixels.")),
        io.Float.Input("frame_rate", default=16.0, min=1.0, step=0.1, display_name="Input FPS",
                       tooltip="Fallback frame rate when the input does not carry fps metadata."),
        *middle,
        io.Float.Input("amount", default=1.0, min=0.0, max=3.0, step=0.05, display_name="Amount", display_mode=slider),
        io.Float.Input("speed", default=1.0, min=0.1, max=3.0, step=0.05, display_name="Speed", display_mode=slider),
        io.Int.Input("seed", default=0, min=0, max=0xFFFFFFFFFFFFFFFF, display_name="Seed",
                     control_after_generate=io.ControlAfterGenerate.fixed),
    ]


def _shake_block(frames_context: Any, frame_rate: float, recipe: ShakeRecipe, amount: float, speed: float, seed: int,
                 node: str, style: str):
    context = hm._normalize_video_input(frames_context)   # only frame count and size are read; pixels are untouched
    block = generate_shake_motion_meta(recipe=recipe, frame_count=len(context.frames), width=context.width,
                                       height=context.height, fps=hm._resolve_fps(context, frame_rate), amount=amount,
                                       speed=speed, seed=seed, node=node, style=style)
    return io.NodeOutput({"motion_meta": block})


class VideoStabilizerShakeGenerator(io.ComfyNode):
    """Deterministic synthetic camera shake by style preset -> motion_meta (host only, no pixels)."""

    @classmethod
    def define_schema(cls) -> io.Schema:
        schema = io.Schema(
            node_id="video_stabilizer_shake_generator",
            display_name="Video Stabilizer Shake Generator",
            category="Video/Stabilization",
            description="Generates deterministic shake motion metadata; it does not alter input frames.",
        )
        schema.inputs = _shake_common_inputs([
            io.Combo.Input("style", options=list(STYLES.keys()), default="handheld", display_name="Style")])
        schema.outputs = [JSONType.Output("motion_meta", display_name="Motion Meta")]
        return schema

    @classmethod
    def execute(cls, frames_context: Any, frame_rate: float, style: str, amount: float, speed: float, seed: int) -> io.NodeOutput:
        return _shake_block(frames_context, frame_rate, STYLES[style], amount, speed, seed, "shake_generator", style)


_MANUAL_FIELDS = [  # id, step, display name (ranges come from shake_generator.FIELD_RANGE, defaults from the handheld preset)
    ("pan", 0.01, "Pan"), ("tilt", 0.01, "Tilt"), ("roll", 0.01, "Roll"), ("zoom", 0.001, "Zoom"),
    ("drift_freq", 0.05, "Drift Frequency"), ("tremor", 0.05, "Tremor"), ("tremor_freq", 0.5, "Tremor Frequency"),
    ("jitter_rate", 0.1, "Jitter Rate"), ("step", 0.05, "Step"), ("randomness", 0.05, "Randomness"),
    ("virtual_fov", 1.0, "Virtual FOV"),
]


class VideoStabilizerShakeGeneratorManual(io.ComfyNode):
    """The same generator driven by explicit recipe values."""

    @classmethod
    def define_schema(cls) -> io.Schema:
        from .shake_generator import FIELD_RANGE

        schema = io.Schema(
            node_id="video_stabilizer_shake_generator_manual",
            display_name="Video Stabilizer Shake Generator Manual",
            category="Video/Stabilization",
            description="Generates deterministic shake motion metadata from manual absolute values.",
        )
        base = STYLES["handheld"]
        fields = []
        for name, step, label in _MANUAL_FIELDS:
            extra = {"display_mode": io.NumberDisplay.slider} if name == "randomness" else {}
            fields.append(io.Float.Input(name, default=getattr(base, name), min=FIELD_RANGE[name][0], max=FIELD_RANGE[name][1],
                                         step=step, display_name=label, **extra))
        schema.inputs = _shake_common_inputs(fields)
        schema.outputs = [JSONType.Output("motion_meta", display_name="Motion Meta")]
        return schema

    @classmethod
    def execute(cls, frames_context: Any, frame_rate: float, pan: float, tilt: float, roll: float, zoom: float,
                drift_freq: float, tremor: float, tremor_freq: float, jitter_rate: float, step: float, randomness: float,
                virtual_fov: float, amount: float, speed: float, seed: int) -> io.NodeOutput:
        recipe = ShakeRecipe(pan, tilt, roll, zoom, drift_freq, tremor, tremor_freq, jitter_rate, step, randomness, virtual_fov)
        return _shake_block(frames_context, frame_rate, recipe, amount, speed, seed, "shake_generator_manual", "manual")


NODE_CLASSES = [VideoStabilizerClassic, VideoStabilizerFlow, VideoStabilizerMotionApply, VideoStabilizerShakeGenerator,
                VideoStabilizerShakeGeneratorManual, VideoStabilizerInverse]


class VideoStabilizerAmdExtension(ComfyExtension):
    async def get_node_list(self) -> list:
        return list(NODE_CLASSES)

    async def on_load(self) -> None:
        """Graph migration Inverse -> Motion Apply, as nodes/node_replacements.py:8-27 registers it (only inside
        a ComfyUI that exposes the node-replacement API)."""
        try:
            from comfy_api.latest import ComfyAPI  # type: ignore
        except ImportError:
            return
        api = ComfyAPI()
        await api.node_replacement.register(
            io.NodeReplace(
                new_node_id="video_stabilizer_motion_apply",
                old_node_id="video_stabilizer_inverse",
                old_widget_ids=["padding_color"],
                input_mapping=[
                    {"new_id": "frames", "old_id": "frames"},
                    {"new_id": "motion_meta", "old_id": "meta"},
                    {"new_id": "padding_color", "old_id": "padding_color"},
                    {"new_id": "framing_mode", "set_value": "crop_and_pad"},
                    {"new_id": "interpolation", "set_value": "bilinear"},
                ],
                output_mapping=[{"new_idx": 0, "old_idx": 0}, {"new_idx": 1, "old_idx": 1}, {"new_idx": 2, "old_idx": 2}],
            )
        )
