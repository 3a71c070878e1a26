"""Whole-clip pipeline of the `Video Stabilizer Flow` node on one MI355X.

Same stages, meta keys and soft-failure rules as the reference's
nodes/video_stabilizer_flow.py:213-640 (`_stabilize_frames`), but batched over the clip:

    gray+downscale (HIP) -> DIS flow for all N-1 pairs (HIP) -> model fit for all pairs (HIP)
    -> sticky-mode selection + parameter deltas (host, sequential by definition)
    -> prefix sum / box smoothing / blend (HIP, fp64) -> framing geometry (host fp64)
    -> warp + padding mask + per-frame padding count for all N frames (HIP)

The host keeps only what is sequential or scalar in the reference (flow.py:324-346, 360-546,
596-640).  There is no CPU fallback for the pixel stages.
"""

from __future__ import annotations

import gc
import os
from dataclasses import dataclass
from typing import Any, Dict, Optional, Tuple

import numpy as np

from . import host_math as hm
from . import native
from .comfy_compat import ProgressBar, check_interrupt
from .meta_v2 import applied_motion_meta_from_arrays, applied_motion_meta_from_stabilization_warp

SAMPLE_STEP = 8  # flow.py:138


class _gc_paused:
    """The plan / meta builders create tens of thousands of small Python objects per clip; a generational GC
    pass landing in the middle costs more than the work itself.  Collection is only postponed, not skipped."""

    def __enter__(self):
        self.was = gc.isenabled()
        gc.disable()

    def __exit__(self, *exc):
        if self.was:
            gc.enable()
        return False


# The Classic node (nodes/video_stabilizer_classic.py) shares this pipeline; only the estimator and a few
# meta keys differ: no flow_backend / flow_fallback_reason (classic.py:189-208, 236-250, 336-360, 537-568),
# no per-transition residual (classic.py:549-557), source "estimated_classic" (classic.py:57-61).
#
# "flow_phase_correlate" is the Flow node running on its fallback estimator (flow.py:90-130).  The reference takes
# that branch when cv2.DISOpticalFlow cannot be created and cv2.optflow (TV-L1, contrib) is missing; here DIS
# always exists, so the branch is selected only by VSTAB_FLOW_BACKEND=phase_correlate (see resolve_flow_backend).
_META_SOURCE = {"flow": "estimated_flow", "flow_phase_correlate": "estimated_flow", "classic": "estimated_classic"}
_PHASE_REASON = "DIS unavailable (disabled by VSTAB_FLOW_BACKEND); cv2.optflow missing; using phase correlation."


def resolve_flow_backend(estimator: str) -> str:
    """_select_flow_backend (flow.py:90-107) for this build: DIS unless the environment asks for the fallback."""
    if estimator != "flow":
        return estimator
    want = os.environ.get("VSTAB_FLOW_BACKEND", "DIS").strip()
    if want in ("", "DIS", "dis"):
        return "flow"
    if want == "phase_correlate":
        return "flow_phase_correlate"
    raise ValueError(f"VSTAB_FLOW_BACKEND={want!r}: expected 'DIS' or 'phase_correlate' (TV-L1 needs opencv-contrib in the "
                     "reference and is not provided).")


def _backend_fields(estimator: str) -> Dict[str, Any]:
    if estimator == "flow":
        return {"flow_backend": "DIS", "flow_fallback_reason": None}
    if estimator == "flow_phase_correlate":
        return {"flow_backend": "phase_correlate", "flow_fallback_reason": _PHASE_REASON}
    return {}


def _attach_motion_meta(meta: Dict[str, Any], fps: float, estimator: str = "flow") -> Dict[str, Any]:
    """flow.py:62-73: a failure to derive motion_meta is swallowed, the rest of the meta survives."""
    try:
        meta["motion_meta"] = applied_motion_meta_from_stabilization_warp(
            meta["stabilization_warp"], fps=fps, source=_META_SOURCE[estimator])
    except (KeyError, TypeError, ValueError, np.linalg.LinAlgError):
        pass
    return meta


def _replay_progress(pbar, done: int, count: int, total: int, stride: int = 10) -> int:
    """Emit the update_absolute sequence the reference's per-item loop produces (flow.py:347-351, 589-593):
    one call per `stride` finished items and one for the remainder."""
    for upto in range(stride, count + 1, stride):
        pbar.update_absolute(done + upto, total)
    if count % stride:
        pbar.update_absolute(done + count, total)
    return done + count


_MODE_INDEX = {"translation": 0, "similarity": 1, "perspective": 2}
_MODE_NAME = ("translation", "similarity", "perspective")


def select_transitions(fit_records, requested_mode: str):
    """Sequential 'sticky active_mode' walk over the per-pair candidate fits (flow.py:324-339, 153-210).

    `fit_records`: structured table [P,3] (native.FIT_DTYPE) or the equivalent list of dicts.
    Returns (matrices f32 [P,3,3] @ working resolution, modes, confidences, residuals, final active mode)."""
    table = fit_records if isinstance(fit_records, np.ndarray) else native.fit_table_from_dicts(fit_records)
    pairs = table.shape[0]
    usable = (table["computed"] != 0) & (table["accepted"] != 0)   # [P,3], mode index 0..2
    active = _MODE_INDEX[requested_mode]
    if pairs and usable[:, active].all():   # the common case: the requested model was accepted for every pair
        rows = table[:, active]
        return (rows["matrix"].reshape(pairs, 3, 3).astype(np.float32), [requested_mode] * pairs, rows["confidence"].tolist(),
                rows["residual"].tolist(), requested_mode)
    chosen = np.empty(pairs, dtype=np.int64)
    p = 0
    while p < pairs:
        # the active mode keeps winning until the first pair whose fit at that mode was rejected
        ok = usable[p:, active]
        run = int(ok.size if ok.all() else np.argmin(ok))
        chosen[p:p + run] = active
        p += run
        if p >= pairs:
            break
        pick = -1
        for m in range(active - 1, -1, -1):   # perspective -> similarity -> translation
            if usable[p, m]:
                pick = m
                break
        chosen[p] = pick
        # no candidate at all (fewer than 12 finite samples, flow.py:153-154): identity, reported as "translation"
        active = pick if pick >= 0 else 0
        p += 1
    idx = chosen
    safe = np.where(idx >= 0, idx, 0)
    rows = table[np.arange(pairs), safe]
    mats = rows["matrix"].reshape(pairs, 3, 3).astype(np.float32)
    confs = rows["confidence"].astype(np.float64)
    resids = rows["residual"].astype(np.float64)
    missing = idx < 0
    if missing.any():
        mats[missing] = np.eye(3, dtype=np.float32)
        confs[missing] = 0.0
        resids[missing] = 0.0
    modes = [_MODE_NAME[m] if m >= 0 else "translation" for m in chosen.tolist()]
    return mats, modes, confs.tolist(), resids.tolist(), _MODE_NAME[active]


def _gray(ctx, device_frames, working_size, peaks_out):
    """F2, plus the per-frame maxima of the same pass when the caller still owes the value-range sniff (F0)."""
    if peaks_out is None:
        return ctx.gray_downscale(device_frames, working_size)
    gray, peaks = ctx.gray_downscale(device_frames, working_size, want_range=True)
    peaks_out.append(hm.prefetch_peaks(peaks))
    return gray


def estimate_transitions(ctx, device_frames, working_size, transform_mode: str, clip_start: bool = True, peaks_out=None):
    """F2-F5 for frames [N,H,W,3] on the device -> per-pair candidate fits (structured table [N-1,3]).
    peaks_out: a list that receives the device tensor of per-frame maxima (see host_math.resolve_value_range)."""
    gray = _gray(ctx, device_frames, working_size, peaks_out)
    _, grid = ctx.dis_flow_batch(gray, sample_step=SAMPLE_STEP, want_full=False, want_grid=True, clip_start=clip_start)
    return ctx.sample_fit_batch(grid, SAMPLE_STEP, transform_mode)


def estimate_transitions_phase(ctx, device_frames, working_size, transform_mode: str, clip_start: bool = True, peaks_out=None):
    """Fallback estimator (flow.py:110-130, 325-330): phase correlation of consecutive estimation images.  Every pair
    is reported as a "translation" fit whatever `transform_mode` asks for; confidence = peak response, residual 0."""
    gray = _gray(ctx, device_frames, working_size, peaks_out)
    table, _ = ctx.phase_correlate_batch(gray)
    return table


_ESTIMATORS = {}   # filled below the three estimator functions


# classic.py:76-96: cv2.goodFeaturesToTrack / cv2.calcOpticalFlowPyrLK arguments of the Classic node
CLASSIC_GFTT = dict(max_corners=400, quality=0.01, min_distance=7.0, block_size=21)
CLASSIC_LK = dict(win=31, max_level=3, max_count=50, epsilon=0.01)


def estimate_transitions_classic(ctx, device_frames, working_size, transform_mode: str, clip_start: bool = True, peaks_out=None):
    """Classic estimator (classic.py:69-160) for frames [N,H,W,3] on the device -> candidate fits [N-1,3]:
    corners of frame i (HIP) -> pyramidal LK into frame i+1 (HIP) -> model fits on the tracked pairs (HIP)."""
    gray = _gray(ctx, device_frames, working_size, peaks_out)
    corners, counts = ctx.gftt_batch(gray[:-1], **CLASSIC_GFTT)
    pairs = ctx.lk_track_batch(gray, corners, counts, **CLASSIC_LK)
    return ctx.points_fit_batch(pairs, counts, transform_mode)


_ESTIMATORS.update({"flow": estimate_transitions, "flow_phase_correlate": estimate_transitions_phase,
                    "classic": estimate_transitions_classic})


def _fps_fields(context: hm.VideoContext, frame_rate) -> Tuple[float, Optional[float]]:
    cand = frame_rate
    if not isinstance(cand, (int, float)) or not np.isfinite(cand) or cand <= 0.0:
        cfps = context.fps
        cand = cfps if isinstance(cfps, (int, float)) and np.isfinite(cfps) and cfps > 0.0 else 16.0
    effective = float(max(1.0, cand))
    requested = float(frame_rate) if isinstance(frame_rate, (int, float)) and frame_rate > 0.0 else None
    return effective, requested


def _host_frames(context: hm.VideoContext) -> np.ndarray:
    if context.batch is not None:
        hm.resolve_value_range(context)
        return context.batch.detach().cpu().numpy()
    if context.batch_u8 is not None:
        arr = context.batch_u8.numpy().astype(np.float32)
        arr /= 255.0
        return arr
    return np.stack([hm._ensure_rgb(f) for f in context.frames], axis=0)


@dataclass
class FlowPlan:
    """Everything the warp stage and the meta need, derived on the host from the per-pair fits
    (replicated on every rank in multi-GPU runs: deterministic fp64 / integer logic only)."""

    final_matrices: np.ndarray         # float32 [N,3,3] (source -> output), stacked: indexable per frame, no list copies
    output_size: Tuple[int, int]
    meta_head: Dict[str, Any]          # keys of flow.py:598-611 that do not depend on the warp
    framing_meta: Dict[str, Any]
    estimated_motion: Dict[str, Any]
    framing_mode: str
    source_size: Tuple[int, int]
    fps_effective: float
    bypass_meta: Optional[Dict[str, Any]] = None
    estimator: str = "flow"


def plan_stabilization(*args, **kwargs) -> "FlowPlan":
    with _gc_paused():
        return _plan_stabilization(*args, **kwargs)


def _plan_stabilization(ctx, fit_records, size, total_frames, framing_mode, transform_mode, camera_lock, strength, smooth,
                       keep_fov, padding_rgb, fps_effective, fps_requested, estimator: str = "flow") -> FlowPlan:
    """flow.py:324-546 for a whole clip: sticky-mode selection, parameter deltas, trajectory (HIP fp64),
    framing geometry.  `fit_records` covers all N-1 transitions of the clip."""
    width, height = size
    rgb_list = [int(c) for c in padding_rgb]
    base_mode = transform_mode
    working_size = hm._working_estimation_size(width, height)
    work_mats, modes_used, confidences, residuals, active_mode = select_transitions(fit_records, transform_mode)
    # rescale to full resolution + parameter deltas (flow.py:340-346), one library call over the clip
    matrices, delta_params = native.transitions_to_params(work_mats, base_mode, size, working_size)

    # ---- trajectory (F7-F8) --------------------------------------------------
    strength = float(np.clip(strength, 0.0, 1.0))
    smooth = float(np.clip(smooth, 0.0, 1.0))
    path, target_path = ctx.trajectory(delta_params, smooth, fps_effective, strength, bool(camera_lock))
    if camera_lock:
        smooth = max(smooth, 0.85)
    diffs = target_path - path

    keep_fov_clamped = float(np.clip(keep_fov, 0.0, 1.0))
    keep_fov_applied = framing_mode == "crop" and keep_fov_clamped > 1e-6
    stabilization_scale = 1.0

    if framing_mode == "crop":
        if keep_fov_clamped >= 0.9999:  # flow.py:387-429: return the original frames
            meta = {
                "frames": total_frames,
                "note": "keep_fov~=1.0 in crop mode; returning original frames.",
                "transform_mode_requested": transform_mode,
                "transform_mode_applied": "identity",
                "camera_lock": camera_lock,
                "strength": strength,
                "strength_effective": 0.0,
                "smooth": smooth,
                "fps_requested": fps_requested,
                "fps_effective": fps_effective,
                "framing": {
                    "mode": framing_mode,
                    "input_size": list(size),
                    "keep_fov_requested": keep_fov_clamped,
                    "keep_fov_effective": 1.0,
                    "min_content_ratio": 1.0,
                    "padding_color_rgb": rgb_list,
                    "stabilization_scale": 0.0,
                },
                "keep_fov_applied": False,
                **_backend_fields(estimator),
                "stabilization_warp": hm._build_stabilization_warp_meta(
                    source_size=size, output_size=size, framing_mode=framing_mode,
                    applied_matrices=[np.eye(3, dtype=np.float32) for _ in range(total_frames)]),
                "estimated_motion": {
                    "per_transition": [],
                    "path": path.tolist(),
                    "target_path": target_path.tolist(),
                    "target_path_effective": path.tolist(),
                },
                "padding_fraction_mean": 0.0,
                "padding_fraction_max": 0.0,
            }
            return FlowPlan(np.zeros((0, 3, 3), np.float32), size, {}, {}, {}, framing_mode, size, fps_effective, bypass_meta=meta,
                            estimator=estimator)
        # flow.py:431-470: keep_fov solver, then the padding-free refinement
        from .crop_solver import solve_crop

        sol = solve_crop(ctx, base_mode, list(diffs), width, height, keep_fov_clamped,
                         max(0.5, 0.02 * max(width, height)), interrupt_check=check_interrupt)
        crop_solution = sol
        apply_matrices = np.stack([np.asarray(m, dtype=np.float32) for m in sol["apply_matrices"]])
        stabilization_scale = sol["scale"]
    else:
        crop_solution = None
        apply_matrices = native.params_to_matrices(diffs, base_mode)
    output_size = size
    mins, maxs = native.bounding_boxes(apply_matrices, width, height)   # f32 matrices (params_to_matrices / crop solver)
    framing_meta: Dict[str, Any] = {
        "mode": framing_mode,
        "input_size": list(size),
        "padding_color_rgb": rgb_list,
        "min_content_ratio": hm._min_content_ratio(mins, maxs, width, height),
    }
    if framing_mode == "crop":  # flow.py:485-499
        final_matrices = np.stack([np.asarray(m, dtype=np.float32) for m in crop_solution["final_matrices"]])
        framing_meta.update({
            "keep_fov_status": crop_solution["status"],
            "keep_fov_effective": crop_solution["keep_fov_effective"],
            "crop_origin": crop_solution["crop_origin"],
            "crop_size": crop_solution["crop_size"],
            "actual_content_ratio": crop_solution["keep_fov_effective"],
            "stabilization_scale": float(stabilization_scale),
        })
        if keep_fov_applied:
            framing_meta["keep_fov_requested"] = keep_fov_clamped
        if crop_solution["note"]:
            framing_meta["keep_fov_note"] = crop_solution["note"]
    elif framing_mode == "crop_and_pad":  # flow.py:500-529
        x0, y0 = float(np.max(mins[:, 0])), float(np.max(mins[:, 1]))
        x1, y1 = float(np.min(maxs[:, 0])), float(np.min(maxs[:, 1]))
        inter_w, inter_h = max(1.0, x1 - x0), max(1.0, y1 - y0)
        off_x = width * 0.5 - (x0 + x1) * 0.5
        off_y = height * 0.5 - (y0 + y1) * 0.5
        shift = np.array([[1.0, 0.0, off_x], [0.0, 1.0, off_y], [0.0, 0.0, 1.0]], dtype=np.float32)
        final_matrices = np.matmul(shift, apply_matrices)
        framing_meta.update({
            "safe_region_origin": [x0, y0],
            "safe_region_size": [inter_w, inter_h],
            "actual_content_ratio": min(inter_w / width, inter_h / height),
            "center_offset": [off_x, off_y],
        })
    elif framing_mode == "expand":  # flow.py:530-533
        shift, output_size = hm._prepare_expand_transform(mins, maxs)
        final_matrices = np.matmul(shift, apply_matrices)
        framing_meta["expanded_size"] = list(output_size)
    else:
        raise ValueError(f"Unsupported framing_mode {framing_mode!r}; expected 'crop', 'crop_and_pad', or 'expand'.")

    effective_diffs = hm.matrices_to_params(apply_matrices, base_mode) if framing_mode == "crop" else diffs  # flow.py:535-539
    stabilization_scale = float(np.clip(stabilization_scale, 0.0, 1.0))
    effective_target_path = path + effective_diffs
    meta_head = {
        "frames": total_frames,
        "transform_mode_requested": transform_mode,
        "transform_mode_applied": active_mode,
        "camera_lock": camera_lock,
        "strength": strength,
        "strength_effective": strength * stabilization_scale,
        "smooth": smooth,
        "fps_requested": fps_requested,
        "fps_effective": fps_effective,
        "keep_fov_applied": keep_fov_applied,
        "padding_color_rgb": rgb_list,
        **_backend_fields(estimator),
    }
    estimated_motion = {   # arrays; turned into JSON lists by prepare_meta (off the critical path)
        "modes": modes_used, "confidences": confidences, "residuals": residuals, "matrices": matrices,
        "path": path, "target_path": target_path, "target_path_effective": effective_target_path,
    }
    return FlowPlan(final_matrices, output_size, meta_head, framing_meta, estimated_motion, framing_mode, size, fps_effective,
                    estimator=estimator)


def prepare_meta(plan: FlowPlan) -> Dict[str, Any]:
    with _gc_paused():
        return _prepare_meta(plan)


def _prepare_meta(plan: FlowPlan) -> Dict[str, Any]:
    """Everything of flow.py:596-640 that does not depend on the warped pixels (the heavy JSON part:
    stabilization_warp + motion_meta).  Called while the warp kernel is still running."""
    h = plan.meta_head
    em = plan.estimated_motion
    final_stack = plan.final_matrices
    meta = {
        "frames": h["frames"],
        "transform_mode_requested": h["transform_mode_requested"],
        "transform_mode_applied": h["transform_mode_applied"],
        "camera_lock": h["camera_lock"],
        "strength": h["strength"],
        "strength_effective": h["strength_effective"],
        "smooth": h["smooth"],
        "fps_requested": h["fps_requested"],
        "fps_effective": h["fps_effective"],
        "framing": dict(plan.framing_meta),
        "keep_fov_applied": h["keep_fov_applied"],
        "padding_color_rgb": h["padding_color_rgb"],
        **_backend_fields(plan.estimator),
        "stabilization_warp": hm._build_stabilization_warp_meta(
            source_size=plan.source_size, output_size=plan.output_size, framing_mode=plan.framing_mode,
            applied_matrices=final_stack),
        "estimated_motion": {
            "per_transition": [
                {"index": i, "mode": mode, "confidence": conf, "residual": resid, "matrix": mat}
                for i, (mode, conf, resid, mat) in enumerate(zip(em["modes"], em["confidences"], em["residuals"],
                                                                 np.asarray(em["matrices"], dtype=np.float32).tolist()))
            ] if plan.estimator != "classic" else [
                {"index": i, "mode": mode, "confidence": conf, "matrix": mat}
                for i, (mode, conf, mat) in enumerate(zip(em["modes"], em["confidences"],
                                                          np.asarray(em["matrices"], dtype=np.float32).tolist()))
            ],
            "path": em["path"].tolist(),
            "target_path": em["target_path"].tolist(),
            "target_path_effective": em["target_path_effective"].tolist(),
        },
        "padding_fraction_mean": None,
        "padding_fraction_max": None,
    }
    fast = applied_motion_meta_from_arrays(final_stack, plan.source_size, plan.output_size, plan.fps_effective,
                                           _META_SOURCE[plan.estimator])
    if fast is not None:
        meta["motion_meta"] = fast
        return meta
    return _attach_motion_meta(meta, plan.fps_effective, plan.estimator)


def complete_meta(meta: Dict[str, Any], plan: FlowPlan, pad_counts) -> Dict[str, Any]:
    """flow.py:583-588, 596, 636-637: padding statistics from the per-frame padded-pixel counts."""
    counts = np.asarray(pad_counts, dtype=np.int64)
    pixels = np.float32(plan.output_size[0] * plan.output_size[1])
    padded_ratios = (counts.astype(np.float32) / pixels).astype(np.float64)  # mask.mean() in float32 (flow.py:587)
    meta["framing"]["padding_detected"] = bool((counts > 0).any())
    meta["padding_fraction_mean"] = float(np.mean(padded_ratios))
    meta["padding_fraction_max"] = float(np.max(padded_ratios))
    return meta


def finish_meta(plan: FlowPlan, pad_counts) -> Dict[str, Any]:
    return complete_meta(prepare_meta(plan), plan, pad_counts)


# ---- the plan formed on the device, speculatively (csrc/vstab_traj.hip: plan_kernel) ---------------------------------
# Between the last model fit and the first warp the reference runs its sequential host logic (flow.py:324-371, 472-521).
# Formed on the host that is a download of the fits, ~0.15 ms of arithmetic, an upload of the matrices and a launch: ~0.24 ms
# in which the GPU idles (3 % of a C2 step, more of a rank's step in a multi-GPU run).  For the common configuration --
# DIS estimator, crop_and_pad, translation / similarity -- the same logic runs as one small fp64 kernel behind the fit
# kernel and the warp is queued behind it at once.  The host still forms the plan, with the HOST's libm as the reference
# does (it needs the fits for the meta anyway), while the warp runs, then compares its float32 final matrices with the
# device's bit for bit: a frame whose matrix differs (the device's atan2 / log / exp / cos / sin may differ from glibc's in
# the last bit of a double, which survives the float32 cast about once in 1e8 entries) is warped again with the host's.
# What is returned is therefore always the host plan's result.  VSTAB_DEVICE_PLAN=0 switches the speculation off (A/B).
_DEVICE_PLAN_MAX_FRAMES = 4096          # plan_kernel keeps the path [frames, 4] fp64 in LDS (perspective: [frames, 8], half as many)
PLAN_PARAMS = {"translation": 2, "similarity": 4, "perspective": 8}   # parameters per frame of a model's path
_DEVICE_PLAN_MAX_SEGMENTS = 64          # plan_kernel's segment table (PLAN_MAX_SEG in csrc/vstab_traj.hip): ranks of a sharded run


def device_plan_applies(estimator: str, framing_mode: str, transform_mode: str, total_frames: int, segments: int = 1) -> bool:
    """Whether the speculative device plan covers this call.  `segments`: the ranks whose record blocks plan_kernel would
    read from an all-gather's receive buffer (a world beyond its segment table takes the host-plan form, which has no limit).
    What a call did is reported in its result (`StabilizationResult.device_plan`, `stats["device_plan"]` of a sharded call):
    {"used": bool, "mismatched_frames": int} -- there is no module-level record."""
    return (os.environ.get("VSTAB_DEVICE_PLAN", "1") not in ("0", "false", "False") and estimator == "flow"
            and framing_mode in ("crop_and_pad", "expand") and transform_mode in PLAN_PARAMS
            and 2 <= total_frames <= (_DEVICE_PLAN_MAX_FRAMES * 4) // max(PLAN_PARAMS[transform_mode], 4)
            and 1 <= segments <= _DEVICE_PLAN_MAX_SEGMENTS)


def _counts_to_host(counts, mirrored: bool = True) -> np.ndarray:
    """The warp's per-frame padded-pixel counts on the host: from the mirror the library keeps behind the warp kernel (no copy,
    no stream synchronisation of ours) where the tensor carries its fetch handle, else by a transfer."""
    fetch = getattr(counts, "_vstab_fetch", None) if mirrored else None
    return fetch() if fetch is not None else counts.cpu().numpy()


def _rewarp_mismatched(ctx, device_frames, plan, final_dev, dst, mask, counts, padding_rgb) -> int:
    """Frames whose device-plan matrix is not the host plan's, bit for bit, are warped again with the host's."""
    host = np.ascontiguousarray(plan.final_matrices, np.float32)
    bad = np.nonzero((host.view(np.uint32) != np.ascontiguousarray(final_dev).view(np.uint32)).reshape(len(host), -1).any(axis=1))[0]
    # runs of consecutive frames go in one launch each, straight into their slices of the outputs: the one situation with
    # MANY differing frames (an exactly constant rotation path after a sticky fallback) costs one more warp of those
    # frames, not a launch per frame
    if bad.size:
        cuts = np.nonzero(np.diff(bad) > 1)[0] + 1
        for run in np.split(bad, cuts):
            a, b = int(run[0]), int(run[-1]) + 1
            _, _, c2 = ctx.warp_batch(device_frames[a:b], host[a:b], plan.output_size, interp="bilinear",
                                      border=hm.border_value(padding_rgb), want_mask=True, want_count=True, out=dst[a:b],
                                      out_mask=mask[a:b])
            counts[a:b].copy_(c2)
    return int(bad.size)


def _stabilize_with_device_plan(ctx, context, device_frames, working_size, total_frames, framing_mode, transform_mode, camera_lock,
                                strength, smooth, keep_fov, padding_rgb, fps_effective, fps_requested, pbar, progress_total,
                                keep_on_device):
    """F2-F14 with the plan formed on the device (see above).  Returns None when F0 found 0..255 float data: the
    speculative run used the unscaled frames and is discarded; the caller takes the regular path on the rescaled clip."""
    size = (context.width, context.height)
    peaks = [] if context.range_pending else None
    gray = _gray(ctx, device_frames, working_size, peaks)
    _, grid = ctx.dis_flow_batch(gray, sample_step=SAMPLE_STEP, want_full=False, want_grid=True)
    pairs = ctx.sample_fit_batch_begin(grid, SAMPLE_STEP, transform_mode)
    ctx.flow_plan_device(ctx.fit_records_device(), pairs, transform_mode, size, working_size, smooth, fps_effective, strength,
                         bool(camera_lock), warp_frames=total_frames, framing=framing_mode)
    out_size = size
    if framing_mode == "expand":
        # the canvas has to exist before the warp is queued: wait for the plan kernel's region (the plan's download, ~30 us
        # behind the fit kernel on the side stream -- not for the host's own plan, which runs under the warp as before)
        out_size = ctx.expand_canvas(ctx.flow_plan_result(total_frames, PLAN_PARAMS[transform_mode])[3])
        if out_size is None:           # a non-finite region: nothing to speculate on
            ctx.sample_fit_batch_end(pairs)
            return None
    dst, mask, counts = ctx.warp_batch_planned(device_frames, 0, out_size, border=hm.border_value(padding_rgb), want_mask=True,
                                               want_count=True)
    fit_records = ctx.sample_fit_batch_end(pairs)          # waits for the fits only; the plan kernel and the warp run on
    if peaks and hm.resolve_value_range(context, peaks[0], ctx):
        return None
    progress_done = _replay_progress(pbar, 0, total_frames - 1, progress_total)
    check_interrupt()
    plan = plan_stabilization(ctx, fit_records, size, total_frames, framing_mode, transform_mode, camera_lock, strength, smooth,
                              keep_fov, padding_rgb, fps_effective, fps_requested, estimator="flow")
    meta = prepare_meta(plan)                               # host JSON work overlaps the warp kernel
    final_dev = ctx.flow_plan_result(total_frames, PLAN_PARAMS[transform_mode])[0]
    if tuple(plan.output_size) != tuple(out_size):
        # expand: the host's canvas is a pixel wider / taller than the device's (an extent within one ulp of an integer): every
        # frame is warped again onto the host plan's canvas
        dst, mask, counts = ctx.warp_batch(device_frames, plan.final_matrices, plan.output_size, interp="bilinear",
                                           border=hm.border_value(padding_rgb), want_mask=True, want_count=True)
        verdict = {"used": True, "mismatched_frames": total_frames}
        mirror_ok = True               # (the counts of that one warp of all frames are what the library mirrored last)
    else:
        verdict = {"used": True, "mismatched_frames": _rewarp_mismatched(ctx, device_frames, plan, final_dev, dst, mask, counts, padding_rgb)}
        mirror_ok = verdict["mismatched_frames"] == 0
    _replay_progress(pbar, progress_done, total_frames, progress_total)   # (host work that needs no pixel: before the last wait)
    # the warp's counts: mirrored to the host behind the kernel (native.last_pad_counts) -- unless frames were warped again,
    # whose counts went into the device tensor only
    meta = complete_meta(meta, plan, _counts_to_host(counts, mirrored=mirror_ok))
    check_interrupt()
    if keep_on_device:
        return hm.StabilizationResult(dst, mask.unsqueeze(-1), meta, verdict)
    return hm.StabilizationResult(dst.cpu().numpy(), mask.cpu().numpy()[..., np.newaxis], meta, verdict)


def _stabilize_frames(
    context: hm.VideoContext,
    framing_mode: str,
    transform_mode: str,
    camera_lock: bool,
    strength: float,
    smooth: float,
    keep_fov: float,
    padding_rgb: Tuple[int, int, int],
    frame_rate: float,
    *,
    ctx: Optional[native.Context] = None,
    keep_on_device: bool = False,
    estimator: str = "flow",
) -> hm.StabilizationResult:
    """Positional signature of the reference (flow.py:213-223); keyword-only extras select the GPU
    context, keep outputs resident in HBM (multi-GPU sharding lives in distributed.py) or switch the
    motion estimator to the Classic node's sparse tracker (classic.py:163-173, same signature)."""
    if estimator not in _META_SOURCE:
        raise ValueError(f"Unknown estimator {estimator!r}; expected 'flow' or 'classic'.")
    estimator = resolve_flow_backend(estimator)
    total_frames = len(context.frames)
    fps_effective, fps_requested = _fps_fields(context, frame_rate)
    size = (context.width, context.height)
    rgb_list = [int(c) for c in padding_rgb]

    if total_frames == 0:  # unreachable through the node (flow.py:242-273)
        meta = {
            "frames": 0,
            "note": "Empty frame sequence; nothing to stabilise.",
            "transform_mode_requested": transform_mode,
            "transform_mode_applied": "identity",
            "camera_lock": camera_lock,
            "strength": strength,
            "strength_effective": 0.0,
            "smooth": smooth,
            "fps_requested": fps_requested,
            "fps_effective": fps_effective,
            "framing": {"mode": framing_mode, "input_size": list(size), "padding_color_rgb": rgb_list},
            "keep_fov_applied": False,
            "padding_color_rgb": rgb_list,
            **_backend_fields(estimator),
            "stabilization_warp": hm._build_stabilization_warp_meta(
                source_size=size, output_size=size, framing_mode=framing_mode, applied_matrices=[]),
            "estimated_motion": {"per_transition": [], "path": [], "target_path": [], "target_path_effective": []},
            "padding_fraction_mean": 0.0,
            "padding_fraction_max": 0.0,
        }
        return hm.StabilizationResult([], [], _attach_motion_meta(meta, fps_effective, estimator))

    progress_total = max(0, total_frames - 1) + total_frames
    pbar = ProgressBar(progress_total)

    if total_frames == 1:  # flow.py:289-310
        meta = {
            "frames": 1,
            "note": "Single-frame input; bypassed stabilization.",
            "transform_mode": transform_mode,
            "framing_mode": framing_mode,
            **({"keep_fov_applied": False} if estimator != "classic" else {}),   # flow.py:297 only; classic.py:236-250 has no such key
            **_backend_fields(estimator),
            "stabilization_warp": hm._build_stabilization_warp_meta(
                source_size=size, output_size=size, framing_mode=framing_mode,
                applied_matrices=[np.eye(3, dtype=np.float32)]),
            "fps_requested": fps_requested,
            "fps_effective": fps_effective,
        }
        pbar.update_absolute(progress_total, progress_total)
        frames_out = _host_frames(context)
        masks_out = np.zeros((1, context.height, context.width, 1), np.float32)
        return hm.StabilizationResult(frames_out, masks_out, _attach_motion_meta(meta, fps_effective, estimator))

    ctx = ctx or native.default_context()
    device_frames = context.device_batch(ctx)
    working_size = hm._working_estimation_size(context.width, context.height)

    if device_plan_applies(estimator, framing_mode, transform_mode, total_frames):
        done = _stabilize_with_device_plan(ctx, context, device_frames, working_size, total_frames, framing_mode, transform_mode,
                                           camera_lock, strength, smooth, keep_fov, padding_rgb, fps_effective, fps_requested,
                                           pbar, progress_total, keep_on_device)
        if done is not None:
            return done
        device_frames = context.device_batch(ctx)   # F0 rescaled the clip: everything is redone on the rescaled frames below

    # ---- estimation (F2-F5) -------------------------------------------------
    estimate = _ESTIMATORS[estimator]
    peaks = [] if context.range_pending else None
    fit_records = estimate(ctx, device_frames, working_size, transform_mode, peaks_out=peaks)
    if peaks and hm.resolve_value_range(context, peaks[0], ctx):
        # F0 (stabilizer_utils.py:127-131): some frame turned out to be 0..255 float data.  The estimation above ran
        # optimistically on the tensor as given (the gray pass reported the per-frame maxima for free); the frames
        # have been rescaled now, so it is repeated on the rescaled clip.  0..1 input -- the ComfyUI IMAGE contract --
        # never takes this branch.
        device_frames = context.device_batch(ctx)
        fit_records = estimate(ctx, device_frames, working_size, transform_mode)
    progress_done = _replay_progress(pbar, 0, total_frames - 1, progress_total)
    check_interrupt()

    plan = plan_stabilization(ctx, fit_records, size, total_frames, framing_mode, transform_mode, camera_lock, strength,
                              smooth, keep_fov, padding_rgb, fps_effective, fps_requested, estimator=estimator)
    if plan.bypass_meta is not None:  # crop + keep_fov ~ 1 (flow.py:387-429): original frames
        pbar.update_absolute(progress_total, progress_total)
        frames_out = device_frames if keep_on_device else _host_frames(context)
        masks_out = (ctx.torch.zeros((total_frames, context.height, context.width, 1), device=ctx.device)
                     if keep_on_device else np.zeros((total_frames, context.height, context.width, 1), np.float32))
        return hm.StabilizationResult(frames_out, masks_out, _attach_motion_meta(plan.bypass_meta, fps_effective, estimator))

    # ---- warp (F13) ------------------------------------------------------------
    dst, mask, counts = ctx.warp_batch(
        device_frames, plan.final_matrices, plan.output_size, interp="bilinear",
        border=hm.border_value(padding_rgb), want_mask=True, want_count=True)
    meta = prepare_meta(plan)  # host JSON work overlaps the warp kernel
    progress_done = _replay_progress(pbar, progress_done, total_frames, progress_total)
    meta = complete_meta(meta, plan, _counts_to_host(counts))
    check_interrupt()
    verdict = {"used": False, "mismatched_frames": 0}
    if keep_on_device:
        return hm.StabilizationResult(dst, mask.unsqueeze(-1), meta, verdict)
    return hm.StabilizationResult(dst.cpu().numpy(), mask.cpu().numpy()[..., np.newaxis], meta, verdict)
