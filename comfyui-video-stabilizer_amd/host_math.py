"""Host-side (fp64 NumPy) pieces of the stabilization path: input adaptation, the parameter space
used for smoothing, framing geometry and the small JSON helpers.

Mirrors the behaviour of the reference's nodes/stabilizer_utils.py (cited per function); the
pixel work these helpers used to feed into OpenCV now goes to libvstab (native.py).
"""

from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Any, Dict, List, Literal, Optional, Sequence, Tuple

import numpy as np

try:  # torch is the frame container at the node boundary
    import torch
except ImportError:  # pragma: no cover
    torch = None

FramingMode = Literal["crop", "crop_and_pad", "expand"]
TransformMode = Literal["translation", "similarity", "perspective"]

DEFAULT_ESTIMATION_MAX_SIDE = 960  # stabilizer_utils.py:245
DEFAULT_PADDING_RGB = (127, 127, 127)  # stabilizer_utils.py:840


@dataclass
class FrameAdapter:
    """What the first frame looked like on the way in (stabilizer_utils.py:52-60)."""

    dtype: Any
    channel_first: bool
    value_range: str
    origin: str
    squeeze_last_dim: bool


@dataclass
class VideoContext:
    """stabilizer_utils.py:63-72 plus an optional batched container for the GPU path."""

    frames: List[Any]
    adapter: FrameAdapter
    width: int
    height: int
    channels: int
    fps: Optional[float]
    template_kind: str
    template_meta: Dict[str, Any]
    batch: Any = field(default=None, repr=False)  # torch.Tensor [N,H,W,3] f32 0..1 if already assembled
    # True while the per-frame value-range sniff of stabilizer_utils.py:127-131 (max > 1.5 -> /255) is still owed for
    # `batch`: the GPU pipelines get the per-frame maxima from their first pass over the pixels (the gray kernel, or
    # vstab_frame_range) instead of reading the clip once more on the host, and call resolve_value_range().
    range_pending: bool = False
    # a uint8 [N,H,W,3] CPU tensor whose float32 form (stabilizer_utils.py:122-126: astype(float32) / 255) has not been made
    # yet: the GPU pipelines get it from the library (the bytes cross PCIe, the division runs on the device), host-only
    # paths from NumPy (_host_frames)
    batch_u8: Any = field(default=None, repr=False)

    def device_batch(self, ctx):
        """[N,H,W,3] f32 tensor on ctx.device (uploaded once, cached)."""
        t = self.batch
        if t is None and self.batch_u8 is not None:
            t = ctx.upload_u8_as_f32(self.batch_u8)
        if t is None:
            t = torch.from_numpy(np.ascontiguousarray(np.stack(self.frames, axis=0), dtype=np.float32))
        if t.device != ctx.device:
            # CPU tensor (the ComfyUI case): pipelined pageable -> pinned -> HBM upload (native.Context.upload)
            t = ctx.upload(t) if t.device.type == "cpu" else t.to(ctx.device, non_blocking=True)
        self.batch = t.contiguous()
        return self.batch


@dataclass
class StabilizationResult:
    frames: Any
    masks: Any
    meta: Dict[str, Any]
    # what the speculative device plan did in this call ({"used": bool, "mismatched_frames": int}; None where no warp ran:
    # empty / single-frame / bypassed clips) -- flow_pipeline: "the plan formed on the device, speculatively"
    device_plan: Optional[Dict[str, Any]] = None


# --------------------------------------------------------------------------- input adaptation (F0)
def _ensure_rgb(frame: np.ndarray) -> np.ndarray:
    """1 channel -> repeated, >3 channels -> truncated (stabilizer_utils.py:224-233)."""
    if frame.ndim == 2:
        frame = frame[..., np.newaxis]
    c = frame.shape[2]
    if c == 1:
        return np.repeat(frame, 3, axis=2)
    if c > 3:
        return frame[..., :3]
    return frame


def _to_numpy_frame(frame: Any) -> Tuple[np.ndarray, FrameAdapter]:
    """One frame -> HxWxC float32 0..1 (stabilizer_utils.py:96-147)."""
    origin = "numpy"
    if torch is not None and isinstance(frame, torch.Tensor):
        origin = "torch"
        arr = frame.detach().cpu().numpy()
    else:
        arr = np.asarray(frame)
    channel_first = False
    if arr.ndim == 3 and arr.shape[0] in (1, 3, 4) and arr.shape[0] < arr.shape[-1]:
        channel_first = True
        arr = np.moveaxis(arr, 0, -1)
    elif arr.ndim == 4 and arr.shape[0] == 1:
        arr = arr[0]
    squeeze = False
    if arr.ndim == 2:
        arr = arr[..., np.newaxis]
        squeeze = True
    elif arr.ndim == 3 and arr.shape[2] == 1:
        squeeze = True
    dtype = arr.dtype
    if dtype == np.uint8:
        arr = arr.astype(np.float32)
        arr /= 255.0
        value_range = "0_255"
    elif bool(arr.size) and float(arr.max()) > 1.5:
        arr = arr.astype(np.float32)
        arr /= 255.0
        value_range = "0_255"
    else:
        value_range = "0_1"
        if dtype != np.float32 or not arr.flags["C_CONTIGUOUS"]:
            arr = np.ascontiguousarray(arr, dtype=np.float32)
    return arr, FrameAdapter(dtype, channel_first, value_range, origin, squeeze)


def _fast_batch(value: Any):
    """The ComfyUI IMAGE case: a float32 [N,H,W,3] tensor, kept as one batch.  The per-frame range sniff (>1.5 -> /255,
    stabilizer_utils.py:127-131) is deferred (VideoContext.range_pending): the first GPU pass over the pixels reports
    the per-frame maxima, see resolve_value_range()."""
    if torch is None or not isinstance(value, torch.Tensor):
        return None
    if value.ndim != 4 or value.dtype != torch.float32 or value.shape[-1] != 3 or value.shape[0] == 0:
        return None
    n, h, w, _ = value.shape
    if h <= 4:  # tiny heights can trip the reference's per-frame channel-first sniff: take the slow path
        return None
    return value.detach().contiguous()


def _fast_batch_u8(value: Any):
    """A uint8 [N,H,W,3] CPU tensor on a box with a GPU: kept as bytes until a pipeline asks for the device batch
    (VideoContext.batch_u8); everything else takes the per-frame path."""
    if torch is None or not isinstance(value, torch.Tensor):
        return None
    if value.ndim != 4 or value.dtype != torch.uint8 or value.shape[-1] != 3 or value.shape[0] == 0 or value.device.type != "cpu":
        return None
    if value.shape[1] <= 4 or not torch.cuda.is_available():
        return None
    return value.detach().contiguous()


def apply_value_range(batch, peaks, ctx=None):
    """stabilizer_utils.py:127-131 on a whole batch: frames whose maximum exceeds 1.5 are divided by 255 (float32).
    `peaks`: per-frame maxima (tensor on any device; NaN compares False, as `float(arr.max()) > 1.5` does).
    Returns (batch -- a rescaled copy if anything changed, else the same tensor --, value_range of frame 0).
    A device batch is rescaled by the library (vstab_apply_value_range: IEEE float32 division like numpy's; a torch
    division on the GPU need not round the same way), a host batch with NumPy."""
    # the comparison runs on the host (no extra kernel launch in the step); peaks that were prefetched to the host
    # behind the pass that produced them (prefetch_peaks) cost no copy here
    fetch = getattr(peaks, "_vstab_fetch", None)
    host = getattr(peaks, "_vstab_host", None)
    if fetch is not None:                     # the gray pass mirrored them into host memory: wait for that kernel alone
        values = fetch()
        if not (values > 1.5).any():          # the ComfyUI IMAGE contract (0..1): nothing to do, NumPy on 256 floats
            return batch, "0_1"
        big = torch.from_numpy(values > 1.5)
    elif host is not None:
        host[1].synchronize()
        values = host[0].numpy()
        if not (values > 1.5).any():
            return batch, "0_1"
        big = host[0] > 1.5
    else:
        big = peaks.cpu() > 1.5
    if bool(big.any()):
        if batch.device.type == "cpu":
            out = batch.clone()
            arr = out.numpy()
            for i in np.nonzero(big.numpy())[0]:
                arr[i] /= np.float32(255.0)
            batch = out
        else:
            from . import native

            batch = (ctx or native.default_context()).apply_value_range(batch, peaks)
    return batch, ("0_255" if bool(big[0]) else "0_1")


def prefetch_peaks(peaks):
    """Start the device -> host copy of a per-frame maxima tensor on the current stream, into page-locked memory, right
    behind the kernel that fills it; apply_value_range then finds the values on the host (the stream has long passed
    the copy by the time the estimation's own synchronisation returns) instead of paying a blocking copy of its own
    between the estimation and the warp, where the GPU idles."""
    if peaks.device.type == "cpu" or hasattr(peaks, "_vstab_fetch"):   # (the gray pass's maxima reach the host by themselves)
        return peaks
    host = torch.empty(peaks.shape, dtype=peaks.dtype, pin_memory=True)
    host.copy_(peaks, non_blocking=True)
    done = torch.cuda.Event()
    done.record()
    peaks._vstab_host = (host, done)
    return peaks


def resolve_value_range(context: "VideoContext", peaks=None, ctx=None) -> bool:
    """Settle a pending range sniff of context.batch.  `peaks` = per-frame maxima from a GPU pass, or None to compute
    them here with tensor ops (host paths that never reach a kernel: single-frame passthrough, crop bypass on CPU).
    Returns True if frames were rescaled, i.e. whatever was derived from the unscaled pixels must be recomputed."""
    if not context.range_pending:
        return False
    if peaks is None:
        peaks = context.batch.reshape(context.batch.shape[0], -1).amax(dim=1)
    new_batch, vrange = apply_value_range(context.batch, peaks, ctx)
    changed = new_batch is not context.batch
    context.batch = new_batch
    context.adapter.value_range = vrange
    context.range_pending = False
    if changed and context.frames and context.frames[0] is not None:
        host = new_batch if new_batch.device.type == "cpu" else None
        context.frames[:] = [host[i].numpy() if host is not None else None for i in range(len(context.frames))]
    return changed


def _normalize_video_input(value: Any) -> VideoContext:
    """stabilizer_utils.py:150-197."""
    if isinstance(value, dict):
        seq = None
        for key in ("frames", "images", "video"):
            if key in value:
                seq = value[key]
                break
        if seq is None:
            raise ValueError("Video input dictionary must contain 'frames'.")
        kind = "dict"
        tmeta = {k: v for k, v in value.items() if k not in ("frames", "images", "video")}
        fps = tmeta.get("fps")
    else:
        seq, kind, tmeta, fps = value, "sequence", {}, None

    batch = _fast_batch(seq)
    if batch is not None:
        n, h, w, _ = batch.shape
        host = batch if batch.device.type == "cpu" else None
        views = [host[i].numpy() if host is not None else None for i in range(n)]
        adapter = FrameAdapter(np.dtype(np.float32), False, "0_1", "torch", False)
        return VideoContext(views, adapter, int(w), int(h), 3, fps, kind, tmeta, batch=batch, range_pending=True)

    batch_u8 = _fast_batch_u8(seq)
    if batch_u8 is not None:
        n, h, w, _ = batch_u8.shape
        adapter = FrameAdapter(np.dtype(np.uint8), False, "0_255", "torch", False)
        return VideoContext([None] * n, adapter, int(w), int(h), 3, fps, kind, tmeta, batch_u8=batch_u8)

    frames: List[np.ndarray] = []
    first: Optional[FrameAdapter] = None
    for item in seq:
        arr, adapter = _to_numpy_frame(item)
        if first is None:
            first = adapter
        elif adapter.channel_first != first.channel_first or adapter.origin != first.origin:
            raise ValueError("Mixed tensor layouts within the same video sequence are not supported.")
        frames.append(_ensure_rgb(arr))
    if not frames:
        raise ValueError("The input video sequence is empty.")
    h, w, c = frames[0].shape
    return VideoContext(frames, first, int(w), int(h), int(c), fps, kind, tmeta)


def _to_host(t, mask=False, levels=1):
    """Device tensor -> CPU tensor (mask=True: a float32 padding mask, which may cross PCIe as bytes: Context.download; levels: the
    samples S of a motion-blurred mask, 1 for a 0 / 1 mask).  With VSTAB_PINNED_OUTPUT=1 the result lives in page-locked memory and is filled by
    one direct DMA (2-3x the rate of the pageable path, which bounces through a staging buffer); the tensor is an
    ordinary CPU tensor for every consumer, but keeps its pages locked while it lives -- hence opt-in."""
    import os

    if t.device.type == "cpu":
        return t
    if os.environ.get("VSTAB_PINNED_OUTPUT", "0") not in ("", "0", "false", "False"):
        out = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
        out.copy_(t, non_blocking=True)
        torch.cuda.current_stream(t.device).synchronize()
        return out
    # default: an ordinary (pageable) CPU tensor, filled through the library's pinned ring by several host threads while
    # the DMA engine fetches the next chunk (vstab_download) -- 2x+ the rate of t.cpu(), which stages on one thread
    from . import native

    with torch.cuda.device(t.device):
        return native.default_context().download(t, mask=mask, levels=levels)


def _reconstruct_video(frames: Any, context: VideoContext) -> Any:
    """New CPU float32 BHWC tensor, dict inputs get a dict back (stabilizer_utils.py:200-221)."""
    if torch is not None and isinstance(frames, torch.Tensor):
        out = frames if frames.shape[0] else torch.zeros((1, context.height, context.width, 3), dtype=torch.float32)
        out = _to_host(out.to(dtype=torch.float32).contiguous())
    else:
        if isinstance(frames, np.ndarray) and frames.ndim == 4:
            stacked = frames if frames.shape[0] else np.zeros((1, context.height, context.width, 3), np.float32)
        else:
            items = list(frames)
            stacked = np.stack(items, axis=0) if items else np.zeros((1, context.height, context.width, 3), np.float32)
        stacked = np.ascontiguousarray(stacked, dtype=np.float32)
        out = torch.from_numpy(stacked) if torch is not None else stacked
    if context.template_kind == "dict":
        payload = dict(context.template_meta)
        payload["frames"] = out
        return payload
    return out


def _convert_masks_for_output(masks: Any, levels: int = 1) -> Any:
    """(N,h,w,1)|(N,h,w) -> (N,h,w) float32 (stabilizer_utils.py:1055-1077)."""
    if torch is not None and isinstance(masks, torch.Tensor):
        if masks.shape[0] == 0:
            return torch.zeros((1, 1, 1), dtype=torch.float32)
        m = masks[..., 0] if masks.ndim == 4 else masks
        return _to_host(m.to(dtype=torch.float32).contiguous(), mask=True, levels=levels)
    if isinstance(masks, np.ndarray) and masks.ndim in (3, 4):
        stacked = np.zeros((1, 1, 1), np.float32) if not masks.shape[0] else (masks[..., 0] if masks.ndim == 4 else masks)
    else:
        planes = [(m[..., 0] if m.ndim == 3 else m).astype(np.float32) for m in masks]
        stacked = np.stack(planes, axis=0) if planes else np.zeros((1, 1, 1), np.float32)
    stacked = np.ascontiguousarray(stacked, dtype=np.float32)
    return torch.from_numpy(stacked) if torch is not None else stacked


# --------------------------------------------------------------------------- estimation geometry (F1, F6)
def _working_estimation_size(width: int, height: int, max_side: int = DEFAULT_ESTIMATION_MAX_SIDE):
    """Long side capped to 960 px, None when already small enough (stabilizer_utils.py:248-268)."""
    longest = max(int(width), int(height))
    if longest <= max_side:
        return None
    scale = max_side / float(longest)
    sw = max(1, int(round(width * scale)))
    sh = max(1, int(round(height * scale)))
    if sw >= width or sh >= height:
        return None
    return sw, sh


def _resolve_fps(context, frame_rate, default: float = 16.0) -> float:
    """First usable value of: fps carried by the input, the widget, the default (stabilizer_utils.py:75-79)."""
    for candidate in (context.fps, frame_rate, default):
        if isinstance(candidate, (int, float)) and np.isfinite(candidate) and candidate > 0.0:
            return float(candidate)
    return float(default)


def _rescale_transform_to_full(matrix: np.ndarray, source_size, working_size) -> np.ndarray:
    """S^-1 @ M @ S in fp64, stored as float32 (stabilizer_utils.py:279-297)."""
    sx = working_size[0] / float(source_size[0])
    sy = working_size[1] / float(source_size[1])
    down = np.array([[sx, 0.0, 0.0], [0.0, sy, 0.0], [0.0, 0.0, 1.0]], dtype=np.float64)
    up = np.array([[1.0 / sx, 0.0, 0.0], [0.0, 1.0 / sy, 0.0], [0.0, 0.0, 1.0]], dtype=np.float64)
    return (up @ matrix.astype(np.float64) @ down).astype(np.float32)


def rescale_transforms_to_full(stack: np.ndarray, source_size, working_size) -> np.ndarray:
    """Batched `_rescale_transform_to_full`: S and S^-1 are diagonal, so every entry of S^-1 @ M @ S is
    one product chain fl(fl(inv_ii * M_ij) * s_jj) (the other dot-product terms are exact zeros); the
    element-wise form below therefore gives the same bits as the 3x3 matmuls."""
    sx = working_size[0] / float(source_size[0])
    sy = working_size[1] / float(source_size[1])
    down = np.array([sx, sy, 1.0], dtype=np.float64)
    up = np.array([1.0 / sx, 1.0 / sy, 1.0], dtype=np.float64)
    return ((up[None, :, None] * np.asarray(stack, dtype=np.float64)) * down[None, None, :]).astype(np.float32)


# --------------------------------------------------------------------------- parameter space (F6, F9)
def _matrix_to_params(matrix: np.ndarray, base_mode: str) -> np.ndarray:
    """stabilizer_utils.py:300-324."""
    if base_mode == "translation":
        return np.array([matrix[0, 2], matrix[1, 2]], dtype=np.float64)
    if base_mode == "similarity":
        a, c = matrix[0, 0], matrix[1, 0]
        scale = math.sqrt(max(a * a + c * c, 1e-10))
        return np.array([matrix[0, 2], matrix[1, 2], math.atan2(c, a), math.log(scale)], dtype=np.float64)
    m = matrix
    return np.array([m[0, 0] - 1.0, m[0, 1], m[0, 2], m[1, 0], m[1, 1] - 1.0, m[1, 2], m[2, 0], m[2, 1]],
                    dtype=np.float64)


def _params_to_matrix(params: np.ndarray, base_mode: str) -> np.ndarray:
    """stabilizer_utils.py:327-358 (float32 result)."""
    if base_mode == "translation":
        return np.array([[1.0, 0.0, params[0]], [0.0, 1.0, params[1]], [0.0, 0.0, 1.0]], dtype=np.float32)
    if base_mode == "similarity":
        tx, ty, theta, log_scale = params
        s = math.exp(log_scale)
        ct, st = math.cos(theta), math.sin(theta)
        return np.array([[s * ct, -s * st, tx], [s * st, s * ct, ty], [0.0, 0.0, 1.0]], dtype=np.float32)
    p = params
    return np.array([[p[0] + 1.0, p[1], p[2]], [p[3], p[4] + 1.0, p[5]], [p[6], p[7], 1.0]], dtype=np.float32)


# element-wise libm calls (exactly what the per-item helpers use) for whole clips: libvstab's host helper runs the
# process's libm over the array; without the library (host-only use) the same functions are reached through
# np.frompyfunc(math.*) -- both give the bits of the per-item form
def _libm(name: str, a, b=None) -> np.ndarray:
    try:
        from . import native

        return native.host_math(name, a, b)
    except (ImportError, OSError, RuntimeError):
        fn = np.frompyfunc(getattr(math, name), 1 if b is None else 2, 1)
        return (fn(a) if b is None else fn(a, b)).astype(np.float64)


def _SQRT(a): return _libm("sqrt", a)
def _ATAN2(a, b): return _libm("atan2", a, b)
def _LOG(a): return _libm("log", a)
def _EXP(a): return _libm("exp", a)
def _COS(a): return _libm("cos", a)
def _SIN(a): return _libm("sin", a)


def matrices_to_params(stack: np.ndarray, base_mode: str) -> np.ndarray:
    """`_matrix_to_params` for a float32 [N,3,3] stack -> float64 [N,P]; same libm calls, same bits."""
    m = np.asarray(stack)
    n = m.shape[0]
    if base_mode == "translation":
        return np.stack([m[:, 0, 2], m[:, 1, 2]], axis=1).astype(np.float64)
    if base_mode == "similarity":
        a, c = m[:, 0, 0], m[:, 1, 0]
        sq = a * a + c * c                      # float32 arithmetic, as with the float32 scalars of the per-item form
        sq64 = np.where(np.float32(1e-10) > sq, 1e-10, sq.astype(np.float64))
        scale = _SQRT(sq64).astype(np.float64)
        theta = _ATAN2(c.astype(np.float64), a.astype(np.float64)).astype(np.float64)
        return np.stack([m[:, 0, 2].astype(np.float64), m[:, 1, 2].astype(np.float64), theta,
                         _LOG(scale).astype(np.float64)], axis=1) if n else np.zeros((0, 4))
    m64 = m.astype(np.float64) if m.dtype != np.float64 else m
    one = np.float32(1.0) if m.dtype == np.float32 else 1.0
    return np.stack([(m[:, 0, 0] - one).astype(np.float64), m64[:, 0, 1], m64[:, 0, 2], m64[:, 1, 0],
                     (m[:, 1, 1] - one).astype(np.float64), m64[:, 1, 2], m64[:, 2, 0], m64[:, 2, 1]], axis=1)


def params_to_matrices(params: np.ndarray, base_mode: str) -> np.ndarray:
    """`_params_to_matrix` for float64 [N,P] -> float32 [N,3,3]."""
    p = np.asarray(params, dtype=np.float64)
    n = p.shape[0]
    out = np.zeros((n, 3, 3), np.float64)
    out[:, 2, 2] = 1.0
    if base_mode == "translation":
        out[:, 0, 0] = 1.0; out[:, 1, 1] = 1.0
        out[:, 0, 2] = p[:, 0]; out[:, 1, 2] = p[:, 1]
    elif base_mode == "similarity":
        s = _EXP(p[:, 3]).astype(np.float64)
        ct, st = _COS(p[:, 2]).astype(np.float64), _SIN(p[:, 2]).astype(np.float64)
        out[:, 0, 0] = s * ct; out[:, 0, 1] = -s * st; out[:, 0, 2] = p[:, 0]
        out[:, 1, 0] = s * st; out[:, 1, 1] = s * ct; out[:, 1, 2] = p[:, 1]
    else:
        out[:, 0, 0] = p[:, 0] + 1.0; out[:, 0, 1] = p[:, 1]; out[:, 0, 2] = p[:, 2]
        out[:, 1, 0] = p[:, 3]; out[:, 1, 1] = p[:, 4] + 1.0; out[:, 1, 2] = p[:, 5]
        out[:, 2, 0] = p[:, 6]; out[:, 2, 1] = p[:, 7]
    return out.astype(np.float32)


def smoothing_window(smooth: float, fps: float) -> int:
    """Odd box-filter length derived from fps (stabilizer_utils.py:367-374)."""
    fps = float(max(1.0, fps))
    seconds = 3.0 / 16.0 + float(np.clip(smooth, 0.0, 1.0)) * (13.0 / 16.0 - 3.0 / 16.0)
    window = max(3, int(round(seconds * fps)))
    return window + 1 if window % 2 == 0 else window


# --------------------------------------------------------------------------- framing geometry (F10-F12)
def _compute_bounding_boxes(matrices: Sequence[np.ndarray], width: int, height: int):
    """Per-frame min/max of the four warped corners (stabilizer_utils.py:1010-1034)."""
    corners = np.array([[0.0, 0.0, 1.0], [width, 0.0, 1.0], [0.0, height, 1.0], [width, height, 1.0]], dtype=np.float64).T
    mins, maxs = [], []
    for m in matrices:
        pts = m @ corners
        pts /= pts[2, :]
        mins.append([pts[0].min(), pts[1].min()])
        maxs.append([pts[0].max(), pts[1].max()])
    return np.array(mins), np.array(maxs)


def bounding_boxes_batched(stack: np.ndarray, width: int, height: int):
    """`_compute_bounding_boxes` for a stacked [N,3,3] array (one batched matmul)."""
    corners = np.array([[0.0, 0.0, 1.0], [width, 0.0, 1.0], [0.0, height, 1.0], [width, height, 1.0]], dtype=np.float64).T
    pts = np.matmul(stack, corners)
    wq = pts[:, 2, :]
    xs, ys = pts[:, 0, :] / wq, pts[:, 1, :] / wq          # [N,4] each, same quotients as `pts /= pts[2]`
    # min / max of four values: pairwise ufuncs on contiguous columns instead of four strided axis reductions
    x0, x1, x2, x3 = np.ascontiguousarray(xs.T)
    y0, y1, y2, y3 = np.ascontiguousarray(ys.T)
    mins = np.stack([np.minimum(np.minimum(x0, x1), np.minimum(x2, x3)), np.minimum(np.minimum(y0, y1), np.minimum(y2, y3))], axis=1)
    maxs = np.stack([np.maximum(np.maximum(x0, x1), np.maximum(x2, x3)), np.maximum(np.maximum(y0, y1), np.maximum(y2, y3))], axis=1)
    return mins, maxs


def _min_content_ratio(mins: np.ndarray, maxs: np.ndarray, width: int, height: int) -> float:
    """stabilizer_utils.py:1037-1052."""
    iw = max(0.0, np.min(maxs[:, 0]) - np.max(mins[:, 0]))
    ih = max(0.0, np.min(maxs[:, 1]) - np.max(mins[:, 1]))
    if iw <= 0.0 or ih <= 0.0:
        return 1e-6
    return max(1e-6, min(iw / width, ih / height))


def _prepare_expand_transform(mins: np.ndarray, maxs: np.ndarray):
    """Translation + canvas size that keep every frame inside (stabilizer_utils.py:386-406)."""
    x_min, y_min = float(np.min(mins[:, 0])), float(np.min(mins[:, 1]))
    x_max, y_max = float(np.max(maxs[:, 0])), float(np.max(maxs[:, 1]))
    out_w = int(math.ceil(x_max - x_min))
    out_h = int(math.ceil(y_max - y_min))
    shift = np.array([[1.0, 0.0, -x_min], [0.0, 1.0, -y_min], [0.0, 0.0, 1.0]], dtype=np.float32)
    return shift, (max(out_w, 1), max(out_h, 1))


# --------------------------------------------------------------------------- JSON helpers
def _parse_padding_color(value) -> Tuple[int, int, int]:
    """'#RRGGBB' | '#RGB' | 'r,g,b' | 0xRRGGBB -> (r,g,b) (stabilizer_utils.py:843-873)."""
    if isinstance(value, str):
        text = value.strip()
        if "," in text or "/" in text:
            try:
                parts = [int(p) for p in text.replace("/", ",").replace(" ", ",").split(",") if p != ""]
                if len(parts) == 1:
                    parts = parts * 3
                if len(parts) != 3:
                    return DEFAULT_PADDING_RGB
                return tuple(int(np.clip(c, 0, 255)) for c in parts)
            except (TypeError, ValueError):
                return DEFAULT_PADDING_RGB
        digits = text[1:] if text.startswith("#") else text
        if len(digits) == 3:
            digits = "".join(ch * 2 for ch in digits)
        if len(digits) != 6:
            return DEFAULT_PADDING_RGB
        try:
            packed = int(digits, 16)
        except (TypeError, ValueError):
            return DEFAULT_PADDING_RGB
    else:
        try:
            packed = int(value)
        except (TypeError, ValueError):
            return DEFAULT_PADDING_RGB
    packed = int(np.clip(packed, 0, 0xFFFFFF))
    return (packed >> 16) & 0xFF, (packed >> 8) & 0xFF, packed & 0xFF


def border_value(padding_rgb: Sequence[int]) -> np.ndarray:
    """rgb/255 evaluated in float32, as cv2 receives it (flow.py:547-548, motion_apply.py:70-72)."""
    return np.array(padding_rgb, dtype=np.float32) / 255.0


def _build_stabilization_warp_meta(*, source_size, output_size, framing_mode, applied_matrices) -> Dict[str, Any]:
    """stabilizer_utils.py:876-896."""
    return {
        "source_size": [int(source_size[0]), int(source_size[1])],
        "output_size": [int(output_size[0]), int(output_size[1])],
        "framing_mode": framing_mode,
        "matrix_convention": "source_to_stabilized",
        "per_frame": [
            {"index": i, "applied_matrix": m}
            for i, m in enumerate(np.asarray(applied_matrices, dtype=np.float32).reshape(-1, 3, 3).tolist())
        ],
    }
