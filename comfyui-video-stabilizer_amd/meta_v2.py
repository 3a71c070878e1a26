"""motion_meta v2 contract: validation, construction and resolution of per-frame 3x3 transforms.

Behavioural mirror of the reference's nodes/motion_meta.py (same dict layout, same error
messages, same selection rules); host-only fp64 NumPy, nothing here touches pixels.
"""

from __future__ import annotations

from dataclasses import dataclass
from typing import Any, Dict, List, Optional, Sequence, Tuple

import numpy as np

INPUT_TO_OUTPUT = "input_to_output"
SOURCE_TO_STABILIZED = "source_to_stabilized"


@dataclass(frozen=True)
class FrameTransform:
    index: int
    matrix: np.ndarray


@dataclass(frozen=True)
class MotionMeta:
    source: str
    frame_count: int
    fps: float
    input_size: Tuple[int, int]
    output_size: Tuple[int, int]
    per_frame: List[FrameTransform]
    generator: Optional[Dict[str, Any]] = None


def _size_field(owner: str, block: Dict[str, Any], key: str) -> Tuple[int, int]:
    """`[width, height]` of positive integers (motion_meta.py:26-37)."""
    raw = block.get(key)
    if not isinstance(raw, (list, tuple)) or len(raw) != 2:
        raise ValueError(f"{owner}.{key} must be [width, height].")
    try:
        w, h = int(raw[0]), int(raw[1])
    except (TypeError, ValueError) as exc:
        raise ValueError(f"{owner}.{key} must contain integer width/height.") from exc
    if w <= 0 or h <= 0:
        raise ValueError(f"{owner}.{key} must contain positive width/height.")
    return w, h


def _matrix_field(owner: str, entry: Any, position: int, key: str) -> np.ndarray:
    """Finite, invertible 3x3 stored under `key` of per_frame[position] (motion_meta.py:40-59)."""
    where = f"{owner}.per_frame[{position}]"
    if not isinstance(entry, dict):
        raise ValueError(f"{where} must be an object.")
    if entry.get("index") != position:
        raise ValueError(f"{where}.index must be {position}, got {entry.get('index')!r}.")
    if key not in entry:
        raise ValueError(f"{where}.{key} is missing.")
    mat = np.asarray(entry[key], dtype=np.float64)
    if mat.shape != (3, 3):
        raise ValueError(f"{where}.{key} must be 3x3.")
    if not np.isfinite(mat).all():
        raise ValueError(f"{where}.{key} must contain finite numbers.")
    try:
        np.linalg.inv(mat)
    except np.linalg.LinAlgError as exc:
        raise ValueError(f"{where}.{key} is not invertible.") from exc
    return mat


def _matrices_fast(owner: str, entries: list, key: str) -> Optional[np.ndarray]:
    """All entries well-formed -> the stacked fp64 matrices after ONE batched finiteness / invertibility
    check; None when anything is irregular (the caller then walks the entries one by one so that the
    first offending entry produces exactly the reference's error message)."""
    try:
        for pos, entry in enumerate(entries):
            if type(entry) is not dict or entry.get("index") != pos:
                return None
        stack = np.asarray([entry[key] for entry in entries], dtype=np.float64)
    except (KeyError, TypeError, ValueError):
        return None
    if stack.ndim != 3 or stack.shape[1:] != (3, 3) or not np.isfinite(stack).all():
        return None
    try:
        np.linalg.inv(stack)
    except np.linalg.LinAlgError:
        return None
    return stack


def _matrices_checked(owner: str, entries: list, key: str) -> List[np.ndarray]:
    stack = _matrices_fast(owner, entries, key) if entries else None
    if stack is not None:
        return list(stack)
    return [_matrix_field(owner, entry, pos, key) for pos, entry in enumerate(entries)]


def validate_motion_meta(block: Dict[str, Any]) -> None:
    """motion_meta.py:62-100."""
    _validated_matrices(block)


def _validated_matrices(block: Dict[str, Any]) -> List[np.ndarray]:
    """validate_motion_meta, handing back the per-frame matrices it had to form for the check (the parser needs them next: a
    256-frame block's nested lists are walked once instead of twice)."""
    if not isinstance(block, dict):
        raise ValueError("motion_meta must be an object.")
    if block.get("version") != 2:
        raise ValueError(f"motion_meta.version must be 2, got {block.get('version')!r}.")
    if block.get("matrix_convention") != INPUT_TO_OUTPUT:
        raise ValueError(
            "motion_meta.matrix_convention must be 'input_to_output', " f"got {block.get('matrix_convention')!r}."
        )
    source = block.get("source")
    if not isinstance(source, str) or not source:
        raise ValueError("motion_meta.source must be a non-empty string.")
    try:
        count = int(block.get("frame_count"))
    except (TypeError, ValueError) as exc:
        raise ValueError("motion_meta.frame_count must be an integer.") from exc
    if count < 0:
        raise ValueError("motion_meta.frame_count must be non-negative.")
    try:
        fps = float(block.get("fps"))
    except (TypeError, ValueError) as exc:
        raise ValueError("motion_meta.fps must be a positive number.") from exc
    if not np.isfinite(fps) or fps <= 0.0:
        raise ValueError("motion_meta.fps must be a positive number.")
    _size_field("motion_meta", block, "input_size")
    _size_field("motion_meta", block, "output_size")
    entries = block.get("per_frame")
    if not isinstance(entries, list):
        raise ValueError("motion_meta.per_frame must be a list.")
    if len(entries) != count:
        raise ValueError(
            "motion_meta.frame_count mismatch: " f"frame_count is {count}, per_frame has {len(entries)} entry/entries."
        )
    matrices = _matrices_checked("motion_meta", entries, "matrix")
    if source == "generated_shake" and not isinstance(block.get("generator"), dict):
        raise ValueError("motion_meta.generator is required when source is 'generated_shake'.")
    return matrices


def build_motion_meta_v2(*, source: str, frame_count: int, fps: float, input_size: Tuple[int, int],
                         output_size: Tuple[int, int], matrices: Sequence[np.ndarray],
                         generator: Optional[Dict[str, Any]] = None) -> Dict[str, Any]:
    """motion_meta.py:123-152: the JSON block (matrices as nested lists of Python floats)."""
    block: Dict[str, Any] = {
        "version": 2,
        "source": source,
        "frame_count": int(frame_count),
        "fps": float(fps),
        "input_size": [int(input_size[0]), int(input_size[1])],
        "output_size": [int(output_size[0]), int(output_size[1])],
        "matrix_convention": INPUT_TO_OUTPUT,
        "per_frame": [
            {"index": pos, "matrix": mat}
            for pos, mat in enumerate(np.asarray(matrices, dtype=np.float64).reshape(-1, 3, 3).tolist())
        ],
    }
    if generator is not None:
        block["generator"] = dict(generator)
    validate_motion_meta(block)
    return block


def applied_motion_meta_from_arrays(applied_f32: np.ndarray, source_size, output_size, fps: float, source: str) -> Dict[str, Any]:
    """Same block as `applied_motion_meta_from_stabilization_warp(warp_meta, ...)` when warp_meta was itself
    built from the float32 stack `applied_f32` (hm._build_stabilization_warp_meta): the float32 -> Python
    float -> float64 round trip is exact, so the checks (finite, invertible) can run on the array once
    instead of re-parsing N nested lists twice.  Any irregularity defers to the generic path, which raises
    the reference's exact error."""
    stack = np.asarray(applied_f32, dtype=np.float32).reshape(-1, 3, 3).astype(np.float64)
    ok = bool(np.isfinite(stack).all())
    if ok and stack.shape[0]:
        try:
            np.linalg.inv(stack)
        except np.linalg.LinAlgError:
            ok = False
    w0, h0 = int(source_size[0]), int(source_size[1])
    w1, h1 = int(output_size[0]), int(output_size[1])
    fps_f = float(fps)
    if not ok or min(w0, h0, w1, h1) <= 0 or not np.isfinite(fps_f) or fps_f <= 0.0 or not isinstance(source, str) or not source:
        return None
    return {
        "version": 2,
        "source": source,
        "frame_count": int(stack.shape[0]),
        "fps": fps_f,
        "input_size": [w0, h0],
        "output_size": [w1, h1],
        "matrix_convention": INPUT_TO_OUTPUT,
        "per_frame": [{"index": pos, "matrix": mat} for pos, mat in enumerate(stack.tolist())],
    }


def _warp_block_fields(warp_meta: Any):
    if not isinstance(warp_meta, dict):
        raise ValueError("stabilization_warp must be an object.")
    if warp_meta.get("matrix_convention") != SOURCE_TO_STABILIZED:
        raise ValueError(
            "stabilization_warp.matrix_convention must be 'source_to_stabilized', "
            f"got {warp_meta.get('matrix_convention')!r}."
        )
    src = _size_field("stabilization_warp", warp_meta, "source_size")
    dst = _size_field("stabilization_warp", warp_meta, "output_size")
    entries = warp_meta.get("per_frame")
    if not isinstance(entries, list):
        raise ValueError("stabilization_warp.per_frame must be a list.")
    return src, dst, entries


def motion_meta_from_stabilization_warp(warp_meta: Dict[str, Any], fps: float, source: str) -> Optional[Dict[str, Any]]:
    """Inverse motion (stabilized -> source) from the applied warp block (motion_meta.py:155-188)."""
    src, dst, entries = _warp_block_fields(warp_meta)
    inverses = []
    for pos, entry in enumerate(entries):
        mat = _matrix_field("stabilization_warp", entry, pos, "applied_matrix")
        try:
            inverses.append(np.linalg.inv(mat))
        except np.linalg.LinAlgError:
            return None
    return build_motion_meta_v2(source=source, frame_count=len(inverses), fps=fps, input_size=dst, output_size=src,
                                matrices=inverses)


def applied_motion_meta_from_stabilization_warp(warp_meta: Dict[str, Any], fps: float, source: str) -> Dict[str, Any]:
    """As-applied motion (source -> stabilized) from the warp block (motion_meta.py:191-220)."""
    src, dst, entries = _warp_block_fields(warp_meta)
    mats = _matrices_checked("stabilization_warp", entries, "applied_matrix")
    return build_motion_meta_v2(source=source, frame_count=len(mats), fps=fps, input_size=src, output_size=dst,
                                matrices=mats)


def _parse_block(block: Dict[str, Any]) -> MotionMeta:
    frames = [FrameTransform(index=pos, matrix=m) for pos, m in enumerate(_validated_matrices(block))]
    gen = block.get("generator")
    return MotionMeta(
        source=str(block["source"]),
        frame_count=int(block["frame_count"]),
        fps=float(block["fps"]),
        input_size=_size_field("motion_meta", block, "input_size"),
        output_size=_size_field("motion_meta", block, "output_size"),
        per_frame=frames,
        generator=dict(gen) if isinstance(gen, dict) else None,
    )


def resolve_motion_meta(meta: Dict[str, Any]) -> MotionMeta:
    """motion_meta.py:223-235."""
    if not isinstance(meta, dict):
        raise ValueError("meta must be a dictionary containing motion_meta or stabilization_warp.")
    block = meta.get("motion_meta")
    if isinstance(block, dict):
        return _parse_block(block)
    warp = meta.get("stabilization_warp")
    if isinstance(warp, dict):
        inv = motion_meta_from_stabilization_warp(warp, fps=16.0, source="legacy_stabilization")
        if inv is None:
            raise ValueError("stabilization_warp contains a non-invertible applied_matrix.")
        return _parse_block(inv)
    raise ValueError("meta must contain motion_meta or stabilization_warp.")
