"""`Video Stabilizer Motion Apply`: replay per-frame 3x3 motion on a clip (plain or motion-blurred warp).

Same selection, framing and meta rules as the reference's nodes/motion_apply.py:24-429; the
per-frame cv2.warpPerspective loops (:75-202) are one batched HIP launch each
(vstab_warp_batch / vstab_warp_blur_batch).
"""

from __future__ import annotations

from dataclasses import dataclass
from typing import Any, Callable, Dict, List, Optional, Tuple

import numpy as np

from . import host_math as hm
from . import native
from .comfy_compat import check_interrupt
from .meta_v2 import MotionMeta, motion_meta_from_stabilization_warp, resolve_motion_meta

ProgressCallback = Callable[[], None]


@dataclass
class MotionApplyResult:
    frames: Any
    masks: Any
    meta: Dict[str, Any]


def _check_interpolation(interpolation: str) -> str:
    if interpolation in ("bilinear", "bicubic"):
        return interpolation
    raise ValueError(f"Unsupported interpolation {interpolation!r}; expected 'bilinear' or 'bicubic'.")


def _validate_context(context: hm.VideoContext, motion: MotionMeta) -> None:
    """motion_apply.py:32-42."""
    if (context.width, context.height) != motion.input_size:
        raise ValueError(
            "Input frames must match motion_meta.input_size "
            f"{motion.input_size}, got {(context.width, context.height)}."
        )
    if len(context.frames) != motion.frame_count:
        raise ValueError(
            "Frame count mismatch: "
            f"got {len(context.frames)} frame(s), metadata has {motion.frame_count} matrix entry/entries."
        )


def _resolve_motion_for_context(meta: Dict[str, Any], context: hm.VideoContext) -> MotionMeta:
    """Pick the block whose input_size matches the connected frames (motion_apply.py:45-67)."""
    if not isinstance(meta, dict):
        return resolve_motion_meta(meta)
    size = (context.width, context.height)
    block = meta.get("motion_meta")
    if isinstance(block, dict):
        direct = resolve_motion_meta({"motion_meta": block})
        if size == direct.input_size:
            return direct
    warp = meta.get("stabilization_warp")
    if isinstance(warp, dict):
        fps = float(block.get("fps", 16.0)) if isinstance(block, dict) else 16.0
        inverse_block = motion_meta_from_stabilization_warp(warp, fps=fps, source="legacy_stabilization")
        if inverse_block is not None:
            inverse = resolve_motion_meta({"motion_meta": inverse_block})
            if size == inverse.input_size:
                return inverse
    return resolve_motion_meta(meta)


def _tick(cb: Optional[ProgressCallback], times: int) -> None:
    if cb is not None:
        for _ in range(int(times)):
            cb()


def _warp(ctx, device_frames, matrices, output_size, interpolation, padding_rgb, motion_blur, samples, *,
          masks_zero: bool, progress_callback, first: int = 0):
    """motion_apply.py:75-202 for frames [first, first+n) of the clip whose motion is `matrices` (all frames of the
    clip; a single-process call has first == 0 and n == len(matrices)): returns device tensors (frames, masks[n,h,w])."""
    n = device_frames.shape[0]
    total = len(matrices)
    border = hm.border_value(padding_rgb)
    # cooperative cancel (flow.py:275-277 polls per frame; here once per batched launch -- the longest stretch that
    # cannot be interrupted is one launch: ~50 ms for 256 x 1080p bicubic with 17 blur samples).  The progress ticks
    # replayed after the launch are the second delivery point: ComfyUI's progress hook raises from update_absolute.
    check_interrupt()
    if n == 0:   # a rank that owns no frames of the clip (multi-GPU, clip shorter than the world): shapes only
        empty = ctx.torch.empty((0, int(output_size[1]), int(output_size[0]), 3), dtype=ctx.torch.float32, device=ctx.device)
        return empty, empty[..., 0]
    if motion_blur <= 0.0 or samples <= 1:
        m32 = np.stack([np.asarray(m, dtype=np.float32) for m in matrices[first:first + n]])
        dst, mask, _ = ctx.warp_batch(device_frames, m32, output_size, interp=interpolation, border=border,
                                      want_mask=not masks_zero)
        _tick(progress_callback, n)
    else:
        s = int(np.clip(samples, 3, 33))
        m64 = np.stack([np.asarray(m, dtype=np.float64) for m in matrices])
        # the sample matrices of a frame need its neighbour's matrix (motion_apply.py:125-134), never its pixels:
        # the whole table goes down, the library picks rows [first, first+n) (vstab_warp_blur_clip_batch)
        dst, mask = ctx.warp_blur_batch(device_frames, m64, output_size, float(motion_blur), s, interp=interpolation,
                                        border=border, want_mask=not masks_zero, clip_first=first)
        _tick(progress_callback, n * (s if total > 1 else 1))
    if mask is None:
        mask = ctx.torch.zeros((n, int(output_size[1]), int(output_size[0])), dtype=ctx.torch.float32, device=ctx.device)
    return dst, mask


def _common_valid_mask(ctx, input_size, output_size, matrices, first, count, progress_callback) -> np.ndarray:
    """AND over ALL frames of the clip of the nearest-neighbour coverage (motion_apply.py:205-227): coverage only, no
    pixels are read (vstab_common_coverage), so a shard evaluates the whole clip's matrices itself; it ticks for its
    own `count` frames."""
    check_interrupt()
    m32 = np.stack([np.asarray(m, dtype=np.float32) for m in matrices])
    common = ctx.common_coverage(m32, input_size, output_size)
    _tick(progress_callback, count)
    return common


def _center_crop_matrix_from_common(common: np.ndarray, output_size: Tuple[int, int]) -> Optional[np.ndarray]:
    """Largest centred, aspect-preserving crop inside the common mask (motion_apply.py:230-285)."""
    out_w, out_h = output_size
    cx, cy = (out_w - 1) * 0.5, (out_h - 1) * 0.5
    aspect = out_w / float(out_h)

    def crop_dims(scale: float, floor_w: bool):
        cw = max(1.0, out_w / scale) if floor_w else out_w / scale
        ch = cw / aspect
        if ch > out_h:
            ch = out_h / scale
            cw = ch * aspect
        return cw, ch

    def fits(scale: float) -> bool:
        cw, ch = crop_dims(scale, True)
        x0, y0 = int(np.ceil(cx - cw * 0.5)), int(np.ceil(cy - ch * 0.5))
        x1, y1 = int(np.floor(cx + cw * 0.5)), int(np.floor(cy + ch * 0.5))
        if x0 < 0 or y0 < 0 or x1 >= out_w or y1 >= out_h or x1 <= x0 or y1 <= y0:
            return False
        return bool(common[y0:y1 + 1, x0:x1 + 1].all())

    lo, hi = 0.0, 1.0
    if not fits(1.0):
        while hi <= 4.0 and not fits(hi):
            hi *= 1.25
        if hi > 4.0:
            return None
    for _ in range(32):
        mid = max((lo + hi) * 0.5, 1.0)
        if fits(mid):
            hi = mid
        else:
            lo = mid
    scale = float(hi)
    cw, ch = crop_dims(scale, False)
    x0, y0 = cx - cw * 0.5, cy - ch * 0.5
    return np.array([[scale, 0.0, -scale * x0], [0.0, scale, -scale * y0], [0.0, 0.0, 1.0]], dtype=np.float64)


def _expand_matrices(matrices: List[np.ndarray], input_size: Tuple[int, int]):
    """motion_apply.py:288-294."""
    # (one batched matmul each instead of a Python loop over the frames: every rank of a sharded replay evaluates the WHOLE
    # clip's matrices here -- 512 frames of C5 took 3.6 ms per rank and step in front of the blur warp's launch; same bits:
    # tests/test_host_golden.py)
    stack = np.stack([np.asarray(m, dtype=np.float64) for m in matrices])
    mins, maxs = hm.bounding_boxes_batched(stack, input_size[0], input_size[1])
    shift, output_size = hm._prepare_expand_transform(mins, maxs)
    return list(np.matmul(shift, stack)), output_size


def apply_motion_on_device(ctx, device_frames, first: int, motion: MotionMeta, meta: Dict[str, Any], padding_rgb, *,
                           framing_mode: str, interpolation: str, motion_blur: float, motion_blur_samples: int,
                           progress_callback: Optional[ProgressCallback] = None):
    """motion_apply.py:311-428 for frames [first, first + n) of the clip described by `motion` (already resolved and
    validated).  Everything that spans the clip -- the common coverage of `crop`, the bounding box of `expand`, the
    neighbour matrix of a blurred frame -- is a function of the replicated matrices alone, so a shard needs no
    exchange with other ranks.  Returns device tensors (frames [n,h,w,3], masks [n,h,w]) and the meta."""
    matrices = [t.matrix for t in motion.per_frame]
    output_size = motion.output_size
    interpolation = _check_interpolation(interpolation)
    result_meta = dict(meta)
    requested = "crop_and_pad" if framing_mode == "pad" else framing_mode
    effective = requested
    motion_blur = float(np.clip(motion_blur, 0.0, 1.0))
    motion_blur_samples = int(np.clip(motion_blur_samples, 3, 33))
    if requested not in ("crop_and_pad", "crop", "expand"):
        raise ValueError(f"Unsupported framing_mode {framing_mode!r}; expected 'crop_and_pad', 'crop', or 'expand'.")

    kw = dict(progress_callback=progress_callback, first=first)
    if requested == "crop_and_pad":
        frames, masks = _warp(ctx, device_frames, matrices, output_size, interpolation, padding_rgb, motion_blur,
                              motion_blur_samples, masks_zero=False, **kw)
    elif requested == "crop":
        common = _common_valid_mask(ctx, motion.input_size, output_size, matrices, first, device_frames.shape[0], progress_callback)
        crop_matrix = _center_crop_matrix_from_common(common, output_size)
        if crop_matrix is None:
            frames, masks = _warp(ctx, device_frames, matrices, output_size, interpolation, padding_rgb, motion_blur,
                                  motion_blur_samples, masks_zero=False, **kw)
            result_meta["framing_fallback"] = "crop_and_pad"
            effective = "crop_and_pad"
        else:
            cropped = [crop_matrix @ m for m in matrices]
            frames, masks = _warp(ctx, device_frames, cropped, output_size, interpolation, padding_rgb, motion_blur,
                                  motion_blur_samples, masks_zero=True, **kw)
    else:
        expanded, output_size = _expand_matrices(matrices, motion.input_size)
        frames, masks = _warp(ctx, device_frames, expanded, output_size, interpolation, padding_rgb, motion_blur,
                              motion_blur_samples, masks_zero=False, **kw)

    result_meta["motion_apply"] = {
        "input_size": [int(motion.input_size[0]), int(motion.input_size[1])],
        "output_size": [int(output_size[0]), int(output_size[1])],
        "framing_mode": effective,
        "interpolation": interpolation,
        "motion_blur": motion_blur,
        "motion_blur_samples": motion_blur_samples,
        "source": motion.source,
    }
    return frames, masks, result_meta


def apply_motion(
    context: hm.VideoContext,
    meta: Dict[str, Any],
    padding_rgb: Tuple[int, int, int],
    *,
    framing_mode: str = "crop_and_pad",
    interpolation: str = "bilinear",
    motion_blur: float = 0.0,
    motion_blur_samples: int = 9,
    progress_callback: Optional[ProgressCallback] = None,
    ctx: Optional[native.Context] = None,
    keep_on_device: bool = False,
) -> MotionApplyResult:
    """Signature of the reference's apply_motion (motion_apply.py:297-307) plus GPU-context extras."""
    def checked_motion():
        motion = _resolve_motion_for_context(meta, context)
        _validate_context(context, motion)
        _check_interpolation(interpolation)
        if ("crop_and_pad" if framing_mode == "pad" else framing_mode) not in ("crop_and_pad", "crop", "expand"):
            raise ValueError(f"Unsupported framing_mode {framing_mode!r}; expected 'crop_and_pad', 'crop', or 'expand'.")
        return motion

    # Frames that are already on the device (a Flow -> Motion Apply chain with device-resident sockets): the range pass below is
    # launched BEFORE the meta is parsed and validated (0.4 ms of host work for 256 frames, during which the GPU had nothing to
    # do: tools/chain_timeline.py).  Host frames: validation first -- a bad meta must not cost an upload.
    resident = context.batch is not None and getattr(context.batch, "device", None) is not None and context.batch.device.type == "cuda"
    motion = None if resident else checked_motion()
    ctx = ctx or native.default_context()
    device_frames = context.device_batch(ctx)
    kw = dict(framing_mode=framing_mode, interpolation=interpolation, motion_blur=motion_blur, motion_blur_samples=motion_blur_samples)
    if context.range_pending:
        # F0's per-frame range sniff (stabilizer_utils.py:127-131), optimistically, as in the Flow node: the maxima pass
        # (one 6.4 GB read, HBM-bound) runs on a second stream BESIDE the warp of the tensor as given (VALU-bound with
        # motion blur) instead of in front of it; only if a frame turns out to be 0..255 data -- never for a ComfyUI
        # IMAGE -- the frames are rescaled and the warp is repeated (its progress ticks were already emitted once).
        torch = ctx.torch
        main = torch.cuda.current_stream(ctx.device)
        side = ctx.side_stream()
        side.wait_stream(main)
        try:
            with torch.cuda.stream(side):
                peaks = hm.prefetch_peaks(ctx.frame_range(device_frames))
            peaks.record_stream(main)   # allocated from the side stream's pool, consumed on the main stream below
            if motion is None:
                motion = checked_motion()
            frames, masks, result_meta = apply_motion_on_device(ctx, device_frames, 0, motion, meta, padding_rgb,
                                                                progress_callback=progress_callback, **kw)
        finally:
            # also when the validation or the warp raised (a bad meta, VstabError, a cancel delivered by a progress tick): the
            # maxima pass still reads the frames, and the caller is free to drop them as soon as this frame unwinds
            main.wait_stream(side)
        if hm.resolve_value_range(context, peaks, ctx):
            check_interrupt()
            device_frames = context.device_batch(ctx)
            frames, masks, result_meta = apply_motion_on_device(ctx, device_frames, 0, motion, meta, padding_rgb,
                                                                progress_callback=None, **kw)
    else:
        if motion is None:
            motion = checked_motion()
        frames, masks, result_meta = apply_motion_on_device(ctx, device_frames, 0, motion, meta, padding_rgb,
                                                            progress_callback=progress_callback, **kw)
    check_interrupt()
    if keep_on_device:
        return MotionApplyResult(frames, masks.unsqueeze(-1), result_meta)
    return MotionApplyResult(frames.cpu().numpy(), masks.cpu().numpy()[..., np.newaxis], result_meta)
