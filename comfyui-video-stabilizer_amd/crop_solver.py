"""framing_mode="crop": keep_fov solver and padding-free refinement (host logic).

Behavioural mirror of the reference's nodes/stabilizer_utils.py:448-837.  The scalar searches (bisection
over the stabilisation scale, binary search over the crop height on an integral image) stay on the host as
in the reference; the per-frame pixel work they used to do with cv2 (nearest coverage, 3x3 close, bounding
box, AND over frames, 3x3 erode) is one libvstab call for the whole clip (`Context.crop_analysis`).
"""

from __future__ import annotations

import math
from typing import Any, Dict, Sequence, Tuple

import numpy as np

from . import host_math as hm
from . import native


def _largest_aspect_ratio_rectangle(binary_mask: np.ndarray, target_width: int, target_height: int):
    """Largest all-valid crop with the target aspect ratio (stabilizer_utils.py:448-504)."""
    if target_width <= 0 or target_height <= 0:
        return None
    height, width = binary_mask.shape
    aspect = float(target_width) / float(target_height)
    # cv2.integral(mask, sdepth=CV_64F) of a 0/1 mask: every entry is a pixel count, exact in int32 as in fp64
    # (the reference compares the window sums with crop_w * crop_h for equality); int32 halves the memory traffic
    integral = np.zeros((height + 1, width + 1), np.int32)
    np.cumsum(binary_mask > 0, axis=1, dtype=np.int32, out=integral[1:, 1:])
    np.cumsum(integral[1:, 1:], axis=0, out=integral[1:, 1:])

    def find_fit(crop_h: int):
        crop_w = int(math.ceil(aspect * crop_h))
        if crop_h <= 0 or crop_h > height or crop_w > width:
            return None
        sums = (integral[crop_h:, crop_w:] - integral[:-crop_h, crop_w:] - integral[crop_h:, :-crop_w]
                + integral[:-crop_h, :-crop_w])
        matches = sums == crop_w * crop_h
        if not np.any(matches):
            return None
        y0 = int(np.clip(round((height - crop_h) * 0.5), 0, matches.shape[0] - 1))
        x0 = int(np.clip(round((width - crop_w) * 0.5), 0, matches.shape[1] - 1))
        if not matches[y0, x0]:
            y0, x0 = np.unravel_index(int(np.argmax(matches)), matches.shape)
        return int(x0), int(y0)

    low, high = 1, min(height, int(math.floor(width / aspect)))
    best = None
    while low <= high:
        crop_h = (low + high) // 2
        loc = find_fit(crop_h)
        if loc is None:
            high = crop_h - 1
        else:
            best = (loc[0], loc[1], crop_h)
            low = crop_h + 1
    if best is None:
        return None
    x0, y0, crop_h = best
    return float(x0), float(y0), aspect * crop_h, float(crop_h)


def _bbox_candidate(delta_params, base_mode, width, height, scale, safety_margin_px) -> Tuple[float, Dict[str, Any]]:
    """evaluate_bbox_only (stabilizer_utils.py:551-609): crop from the intersection of the warped frame bounds."""
    s = float(np.clip(scale, 0.0, 1.0))
    # one library call each over the clip (vstab_params_to_matrices, vstab_bounding_boxes: the per-item forms' arithmetic, pinned to
    # them by tests/test_abi_cpu.py) instead of two Python loops over the frames per bisection step -- up to nineteen steps
    mats = native.params_to_matrices(np.asarray(delta_params, dtype=np.float64) * s, base_mode)
    mins, maxs = native.bounding_boxes(mats, width, height)
    x0, y0 = float(np.max(mins[:, 0])), float(np.max(mins[:, 1]))
    x1, y1 = float(np.min(maxs[:, 0])), float(np.min(maxs[:, 1]))
    safe_w, safe_h = max(0.0, x1 - x0), max(0.0, y1 - y0)
    margin = min(safety_margin_px, safe_w * 0.25, safe_h * 0.25)
    safe_x0, safe_y0 = x0 + margin, y0 + margin
    safe_w, safe_h = max(0.0, safe_w - 2.0 * margin), max(0.0, safe_h - 2.0 * margin)
    if safe_w <= 0.0 or safe_h <= 0.0:
        return 0.0, {"scale": scale, "pre_crop": mats, "final": mats, "crop_origin": [0.0, 0.0],
                     "crop_size": [float(width), float(height)], "has_overlap": False}
    ratio = min(1.0, safe_w / width, safe_h / height)
    crop_w, crop_h = width * ratio, height * ratio
    cx0 = safe_x0 + (safe_w - crop_w) * 0.5
    cy0 = safe_y0 + (safe_h - crop_h) * 0.5
    cs = width / crop_w
    crop = np.array([[cs, 0.0, -cs * cx0], [0.0, cs, -cs * cy0], [0.0, 0.0, 1.0]], dtype=np.float32)
    return ratio, {"scale": scale, "pre_crop": mats, "final": np.matmul(crop, mats), "crop_origin": [cx0, cy0],
                   "crop_size": [crop_w, crop_h], "has_overlap": True}


def _ratio_from_bboxes(bbox: np.ndarray, width: int, height: int) -> float:
    """min over frames of min(w/W, h/H) of the closed coverage's bounding box (stabilizer_utils.py:632-646)."""
    min_ratio = 1.0
    for x_min, y_min, x_max, y_max in bbox:
        if x_max < 0:
            ratio = 0.0
        else:
            size = [float(max(1, x_max - x_min + 1)), float(max(1, y_max - y_min + 1))]
            ratio = min(size[0] / width, size[1] / height)
        if ratio < min_ratio:
            min_ratio = ratio
    return float(min_ratio)


def solve_crop(ctx, base_mode: str, delta_params: Sequence[np.ndarray], width: int, height: int, keep_fov_target: float,
               safety_margin_px: float, max_iterations: int = 18, interrupt_check=None):
    """`_compute_crop_with_keep_fov_parametric(..., return_masks=False)` followed by
    `_refine_no_padding_crop(..., safety_shrink_px=1)` exactly as flow.py:431-470 chains them.

    Returns dict(final_matrices, apply_matrices, keep_fov_effective, status, note, scale, crop_origin, crop_size)."""
    keep = float(np.clip(keep_fov_target, 0.0, 1.0))
    eps = 1e-4
    size = (width, height)

    def evaluate(scale):
        return _bbox_candidate(delta_params, base_mode, width, height, scale, safety_margin_px)

    ratio_full, raw_full = evaluate(1.0)
    if keep <= eps:
        if bool(raw_full["has_overlap"]):
            chosen, status, note, scale = raw_full, "disabled", None, 1.0
        else:
            _, chosen = evaluate(0.0)
            status, scale = "disabled", 0.0
            note = "No common crop region at full stabilization; stabilization was disabled."
        pending_status = None
    elif ratio_full >= keep - eps:
        chosen, status, note, scale, pending_status = raw_full, "met", None, 1.0, None
    else:
        low, high = 0.0, 1.0
        best = None
        for _ in range(max_iterations):
            mid = 0.5 * (low + high)
            ratio_mid, raw_mid = evaluate(mid)
            if ratio_mid >= keep - eps:
                best, low = raw_mid, mid
            else:
                high = mid
        if best is None:
            _, chosen = evaluate(0.0)
            status = "failed" if keep > eps else "disabled"
            note = None if keep <= eps else f"keep_fov target {keep:.3f} could not be satisfied even with zero stabilisation."
            scale, pending_status = 0.0, None
        else:
            chosen, scale, status, note, pending_status = best, float(best["scale"]), None, None, "search"

    if interrupt_check is not None:
        interrupt_check()
    final = [np.asarray(m, dtype=np.float32) for m in chosen["final"]]
    bbox, common = ctx.crop_analysis(np.stack(final), size, size)
    ratio_final = _ratio_from_bboxes(bbox, width, height)
    if pending_status == "search":
        status = "met" if ratio_final >= keep - eps else "clamped"
        if status == "clamped" and keep > eps:
            note = f"keep_fov target {keep:.3f} reduced to {ratio_final:.3f} at stabilisation scale {scale:.3f}."

    # ---- _refine_no_padding_crop (stabilizer_utils.py:749-837)
    crop_origin, crop_size, keep_effective = [0.0, 0.0], [float(width), float(height)], 0.0
    refined = list(final)
    if common.max() != 0:
        rect = _largest_aspect_ratio_rectangle(common, width, height)
        if rect is not None:
            x0, y0, crop_w, crop_h = rect
            cs = width / crop_w
            crop = np.array([[cs, 0.0, -cs * x0], [0.0, cs, -cs * y0], [0.0, 0.0, 1.0]], dtype=np.float32)
            refined = list(np.matmul(crop, np.stack(final)))
            crop_origin, crop_size, keep_effective = [x0, y0], [crop_w, crop_h], 1.0
    return {
        "final_matrices": refined,
        "apply_matrices": chosen["pre_crop"],
        "keep_fov_effective": keep_effective,
        "status": status,
        "note": note,
        "scale": scale,
        "crop_origin": crop_origin,
        "crop_size": crop_size,
    }
