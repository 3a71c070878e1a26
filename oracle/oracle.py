"""ctypes front-end of the CPU oracle (TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module.  It is the checker, never the product: the product path lives in
comfyui-video-stabilizer_amd/ and talks to libvstab.so (HIP) only.

The C sources restate the OpenCV algorithms behind the reference's call sites
(see oracle/vo_common.h); parity against a real OpenCV is unpinned.
"""

from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
# VSTAB_ORACLE_LIB: another build of the checker (the sanitizer build of `make -C oracle sanitize`, tools/oracle_sanitize.sh)
_LIB_PATH = Path(os.environ["VSTAB_ORACLE_LIB"]).resolve() if os.environ.get("VSTAB_ORACLE_LIB") else _HERE / "_build" / "libvstab_oracle.so"

INTERP = {"bilinear": 0, "bicubic": 1}
SUBPIX = {"q5": 0, "exact": 1}
MODES = {"translation": 0, "similarity": 1, "perspective": 2}
MODE_NAMES = {v: k for k, v in MODES.items()}


def build(force: bool = False) -> Path:
    """Compile the oracle with gcc (a few seconds)."""
    stale = not _LIB_PATH.exists() or any(
        f.stat().st_mtime > _LIB_PATH.stat().st_mtime for pat in ("vo_*.c", "*.h", "Makefile") for f in _HERE.glob(pat))
    if force or stale:
        subprocess.run(["make", "-C", str(_HERE)], check=True, capture_output=True)
    return _LIB_PATH


_lib = None


class DisParams(C.Structure):
    _fields_ = [
        ("finest_scale", C.c_int),
        ("patch_size", C.c_int),
        ("patch_stride", C.c_int),
        ("grad_descent_iter", C.c_int),
        ("var_iter", C.c_int),
        ("alpha", C.c_float),
        ("delta", C.c_float),
        ("gamma", C.c_float),
        ("use_mean_norm", C.c_int),
        ("use_spatial_prop", C.c_int),
    ]


class FitResult(C.Structure):
    _fields_ = [
        ("matrix", C.c_float * 9),
        ("mode", C.c_int),
        ("confidence", C.c_double),
        ("residual", C.c_double),
        ("valid", C.c_int),
    ]


def lib():
    global _lib
    if _lib is None:
        if not _LIB_PATH.exists():
            build()
        _lib = C.CDLL(str(_LIB_PATH))
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _ptr(a, ty):
    return a.ctypes.data_as(C.POINTER(ty))


def set_threads(n: int) -> None:
    """OpenMP threads of the oracle's parallel loops: through the environment before the runtime starts, and through
    omp_set_num_threads once it has (the environment variable is read only once per process)."""
    os.environ["OMP_NUM_THREADS"] = str(int(n))
    try:
        C.CDLL("libgomp.so.1").omp_set_num_threads(int(n))
    except OSError:
        pass


# ---------------------------------------------------------------- warp
def invert3x3(m):
    m = np.ascontiguousarray(m, dtype=np.float64).reshape(9)
    out = np.zeros(9, dtype=np.float64)
    lib().vo_invert3x3(_ptr(m, C.c_double), _ptr(out, C.c_double))
    return out.reshape(3, 3)


def interp_tables():
    lin = np.zeros(64, np.float32)
    cub = np.zeros(128, np.float32)
    lib().vo_interp_tables(_ptr(lin, C.c_float), _ptr(cub, C.c_float))
    return lin.reshape(32, 2), cub.reshape(32, 4)


def warp_frame(src, matrix, out_size, interp="bilinear", border=(0.0, 0.0, 0.0), subpix="q5", want_coverage=True):
    """cv2.warpPerspective(src, matrix(f32), (w,h), flags, BORDER_CONSTANT, border) + nearest coverage."""
    src = _f32(src)
    sh, sw, _ = src.shape
    dw, dh = int(out_size[0]), int(out_size[1])
    m = _f32(matrix).reshape(9)
    b = _f32(border).reshape(3)
    dst = np.empty((dh, dw, 3), np.float32)
    cov = np.empty((dh, dw), np.float32) if want_coverage else None
    lib().vo_warp_frame(
        _ptr(src, C.c_float), sh, sw, _ptr(m, C.c_float), dh, dw, INTERP[interp], _ptr(b, C.c_float),
        SUBPIX[subpix], _ptr(dst, C.c_float), _ptr(cov, C.c_float) if cov is not None else None,
    )
    return dst, cov


def warp_clip(src, matrices, out_size, interp="bilinear", border=(0.0, 0.0, 0.0), subpix="q5", want_mask=True):
    src = _f32(src)
    n, sh, sw, _ = src.shape
    dw, dh = int(out_size[0]), int(out_size[1])
    m = _f32(matrices).reshape(n, 9)
    b = _f32(border).reshape(3)
    dst = np.empty((n, dh, dw, 3), np.float32)
    mask = np.empty((n, dh, dw), np.float32) if want_mask else None
    cnt = np.zeros(n, np.uint32)
    lib().vo_warp_clip(
        _ptr(src, C.c_float), n, sh, sw, _ptr(m, C.c_float), dh, dw, INTERP[interp], _ptr(b, C.c_float),
        SUBPIX[subpix], _ptr(dst, C.c_float), _ptr(mask, C.c_float) if mask is not None else None,
        _ptr(cnt, C.c_uint32),
    )
    return dst, mask, cnt


def warp_blur_clip(src, matrices64, out_size, blur, samples, interp="bilinear", border=(0.0, 0.0, 0.0), subpix="q5", want_mask=True):
    src = _f32(src)
    n, sh, sw, _ = src.shape
    dw, dh = int(out_size[0]), int(out_size[1])
    m = np.ascontiguousarray(matrices64, dtype=np.float64).reshape(n, 9)
    b = _f32(border).reshape(3)
    dst = np.empty((n, dh, dw, 3), np.float32)
    mask = np.empty((n, dh, dw), np.float32) if want_mask else None
    lib().vo_warp_blur_clip.argtypes = [
        C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int,
        C.c_double, C.c_int, C.c_void_p, C.c_void_p,
    ]
    lib().vo_warp_blur_clip(
        src.ctypes.data, n, sh, sw, m.ctypes.data, dh, dw, INTERP[interp], b.ctypes.data, SUBPIX[subpix],
        float(blur), int(samples), dst.ctypes.data, mask.ctypes.data if mask is not None else None,
    )
    return dst, mask


def linspace(a, b, n):
    out = np.zeros(n, np.float64)
    lib().vo_linspace.argtypes = [C.c_double, C.c_double, C.c_int, C.c_void_p]
    lib().vo_linspace(float(a), float(b), int(n), out.ctypes.data)
    return out


# ---------------------------------------------------------------- gray / resize
def rgb2gray_u8(rgb, fused_body=True):
    rgb = _f32(rgb)
    h, w, _ = rgb.shape
    out = np.empty((h, w), np.uint8)
    lib().vo_rgb2gray_u8(_ptr(rgb, C.c_float), h, w, 1 if fused_body else 0, _ptr(out, C.c_uint8))
    return out


def frame_max(frames):
    """Per-frame `float(arr.max())` of stabilizer_utils.py:127-131 for a clip [n,...] f32 (NaN-propagating)."""
    frames = _f32(frames)
    n = frames.shape[0]
    out = np.empty(n, np.float32)
    fn = lib().vo_frame_max
    fn.argtypes = [C.c_void_p, C.c_int, C.c_longlong, C.c_void_p]
    fn.restype = None
    fn(frames.ctypes.data, n, int(frames[0].size), out.ctypes.data)
    return out


def rgb2gray_f32(rgb, fused_body=True):
    """cv2.cvtColor(rgb f32, COLOR_RGB2GRAY) -> f32 [h,w]."""
    rgb = _f32(rgb)
    h, w, _ = rgb.shape
    out = np.empty((h, w), np.float32)
    lib().vo_rgb2gray_f32(_ptr(rgb, C.c_float), h, w, 1 if fused_body else 0, _ptr(out, C.c_float))
    return out


def resize_area_u8(src, out_size):
    src = np.ascontiguousarray(src, dtype=np.uint8)
    sh, sw = src.shape
    dw, dh = int(out_size[0]), int(out_size[1])
    out = np.empty((dh, dw), np.uint8)
    lib().vo_resize_area_u8(_ptr(src, C.c_uint8), sh, sw, _ptr(out, C.c_uint8), dh, dw)
    return out


def resize_linear_f32(src, out_size):
    src = _f32(src)
    if src.ndim == 2:
        src = src[..., None]
    sh, sw, cn = src.shape
    dw, dh = int(out_size[0]), int(out_size[1])
    out = np.empty((dh, dw, cn), np.float32)
    lib().vo_resize_linear_f32(_ptr(src, C.c_float), sh, sw, cn, _ptr(out, C.c_float), dh, dw)
    return out


def gray_for_estimation(frames, work_size, fused_body=True):
    """[_make_gray_for_estimation(f, work_size) for f in frames]; work_size=(w,h) or None."""
    frames = _f32(frames)
    n, h, w, _ = frames.shape
    ww, wh = (w, h) if work_size is None else (int(work_size[0]), int(work_size[1]))
    out = np.empty((n, wh, ww), np.uint8)
    lib().vo_gray_for_estimation(_ptr(frames, C.c_float), n, h, w, wh, ww, 1 if fused_body else 0, _ptr(out, C.c_uint8))
    return out


# ---------------------------------------------------------------- DIS optical flow
def dis_params(**overrides):
    p = DisParams()
    lib().vo_dis_default_params(C.byref(p))
    for k, v in overrides.items():
        setattr(p, k, v)
    return p


def dis_coarsest_scale(h, w, patch_size=8):
    return int(lib().vo_dis_coarsest_scale(int(h), int(w), int(patch_size)))


def dis_flow(i0, i1, params=None):
    """cv2.DISOpticalFlow(...).calc(i0, i1, None) -> [h,w,2] f32."""
    i0 = np.ascontiguousarray(i0, dtype=np.uint8)
    i1 = np.ascontiguousarray(i1, dtype=np.uint8)
    h, w = i0.shape
    p = params or dis_params()
    flow = np.empty((h, w, 2), np.float32)
    rc = lib().vo_dis_calc(_ptr(i0, C.c_uint8), _ptr(i1, C.c_uint8), h, w, C.byref(p), _ptr(flow, C.c_float))
    if rc != 0:
        raise ValueError(f"vo_dis_calc: unsupported configuration (rc={rc})")
    return flow


def dis_patch_search(i0, i1, u0=None, v0=None, params=None):
    """The patch inverse search of ONE level (test-only view): level images i0 / i1 [h,w] u8, dense initial flow (default 0)
    -> sparse flow (sx, sy) [hs,ws] of the 8x8 patches at stride 4."""
    i0 = np.ascontiguousarray(i0, np.uint8)
    i1 = np.ascontiguousarray(i1, np.uint8)
    h, w = i0.shape
    p = params or dis_params()
    u0 = np.zeros((h, w), np.float32) if u0 is None else np.ascontiguousarray(u0, np.float32)
    v0 = np.zeros((h, w), np.float32) if v0 is None else np.ascontiguousarray(v0, np.float32)
    hs, ws = 1 + (h - p.patch_size) // p.patch_stride, 1 + (w - p.patch_size) // p.patch_stride
    sx, sy = np.zeros((hs, ws), np.float32), np.zeros((hs, ws), np.float32)
    rc = lib().vo_dis_patch_search_debug(_ptr(i0, C.c_uint8), _ptr(i1, C.c_uint8), h, w, _ptr(u0, C.c_float), _ptr(v0, C.c_float),
                                         _ptr(sx, C.c_float), _ptr(sy, C.c_float), C.byref(p))
    if rc != 0:
        raise ValueError("vo_dis_patch_search_debug: bad arguments")
    return sx, sy


def variational_refine(i0, i1, u, v, params=None, sor_iters=5):
    """The variational refinement (OpenCV's VariationalRefinement::calcUV as DIS configures it) of flow (u, v) on one level
    (test-only view): returns the refined (u, v).  sor_iters: SOR iterations per fixed-point iteration (DIS uses 5)."""
    i0 = np.ascontiguousarray(i0, np.uint8)
    i1 = np.ascontiguousarray(i1, np.uint8)
    h, w = i0.shape
    uu, vv = np.array(u, np.float32, copy=True, order="C"), np.array(v, np.float32, copy=True, order="C")
    p = params or dis_params()
    rc = lib().vo_variational_refine_debug(_ptr(i0, C.c_uint8), _ptr(i1, C.c_uint8), h, w, _ptr(uu, C.c_float), _ptr(vv, C.c_float),
                                           C.byref(p), int(sor_iters))
    if rc != 0:
        raise ValueError("vo_variational_refine_debug: bad arguments")
    return uu, vv


def dis_gradients(img, patch_size=8, patch_stride=4):
    """Sobel gradients (s16) and structure-tensor planes [5,hs,ws] of one DIS level image (test-only view)."""
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    ix = np.empty((h, w), np.int16)
    iy = np.empty((h, w), np.int16)
    hs, ws = 1 + (h - patch_size) // patch_stride, 1 + (w - patch_size) // patch_stride
    tensor = np.empty((5, hs, ws), np.float32)
    lib().vo_dis_gradients(_ptr(img, C.c_uint8), h, w, patch_size, patch_stride, _ptr(ix, C.c_int16), _ptr(iy, C.c_int16),
                           _ptr(tensor, C.c_float))
    return ix, iy, tensor


class dis_sum_order:
    """with dis_sum_order("butterfly"): ... -- the four patch sums of the inverse search use the XOR butterfly that
    round 1 shipped instead of OpenCV's SIMD128 row accumulators (the default, which the HIP kernel reproduces);
    deviation measurement only (vo_dis.c)."""

    def __init__(self, order):
        self.order = {"butterfly": 0, "opencv": 1}[order]

    def __enter__(self):
        self.prev = lib().vo_dis_get_sum_order()
        lib().vo_dis_set_sum_order(self.order)

    def __exit__(self, *exc):
        lib().vo_dis_set_sum_order(self.prev)
        return False


def dis_flow_stateful(i0, i1, params):
    """calc() on a persistent DIS object: `params` (DisParams) is updated in place like OpenCV's object state."""
    i0 = np.ascontiguousarray(i0, dtype=np.uint8)
    i1 = np.ascontiguousarray(i1, dtype=np.uint8)
    h, w = i0.shape
    flow = np.empty((h, w, 2), np.float32)
    rc = lib().vo_dis_calc_stateful(_ptr(i0, C.c_uint8), _ptr(i1, C.c_uint8), h, w, C.byref(params), _ptr(flow, C.c_float))
    if rc != 0:
        raise ValueError(f"vo_dis_calc_stateful: unsupported configuration (rc={rc})")
    return flow


def warp_frame_inv(src, inverse64, out_size, interp="bilinear", border=(0.0, 0.0, 0.0), subpix="q5"):
    """The warp for a matrix already converted to f64 and inverted (cv2.warpPerspective's internal state after
    `M.convertTo(CV_64F); invert(M)`): serves callers that pass float64 matrices."""
    src = _f32(src)
    sh, sw, _ = src.shape
    dw, dh = int(out_size[0]), int(out_size[1])
    inv = np.ascontiguousarray(inverse64, dtype=np.float64).reshape(9)
    b = _f32(border).reshape(3)
    dst = np.empty((dh, dw, 3), np.float32)
    cov = np.empty((dh, dw), np.float32)
    lib().vo_warp_frame_inv(_ptr(src, C.c_float), sh, sw, _ptr(inv, C.c_double), dh, dw, INTERP[interp], _ptr(b, C.c_float),
                            SUBPIX[subpix], _ptr(dst, C.c_float), _ptr(cov, C.c_float))
    return dst, cov


def dis_flow_clip(gray, params=None):
    gray = np.ascontiguousarray(gray, dtype=np.uint8)
    n, h, w = gray.shape
    p = params or dis_params()
    flow = np.empty((n - 1, h, w, 2), np.float32)
    rc = lib().vo_dis_calc_clip(_ptr(gray, C.c_uint8), n, h, w, C.byref(p), _ptr(flow, C.c_float))
    if rc != 0:
        raise ValueError(f"vo_dis_calc_clip: unsupported configuration (rc={rc})")
    return flow


# ---------------------------------------------------------------- model fit
def fit_all_modes(flow, step=8, requested_mode="similarity"):
    """All candidate models at or below `requested_mode` for one flow field [h,w,2]."""
    flow = _f32(flow)
    h, w, _ = flow.shape
    recs = (FitResult * 3)()
    nv, nt = C.c_int(), C.c_int()
    lib().vo_fit_all_modes(_ptr(flow, C.c_float), h, w, int(step), MODES[requested_mode], recs, C.byref(nv), C.byref(nt))
    out = {}
    for mi in range(3):
        r = recs[mi]
        if r.mode < 0:
            continue
        out[MODE_NAMES[mi]] = {
            "matrix": np.array(list(r.matrix), np.float32).reshape(3, 3),
            "confidence": float(r.confidence),
            "residual": float(r.residual),
            "accepted": bool(r.valid),
        }
    return out, int(nv.value), int(nt.value)


def fit_from_flow(flow, step=8, requested_mode="similarity"):
    """_estimate_motion_flow's tail (flow.py:141-210): (matrix f32 3x3, mode, confidence, residual)."""
    flow = _f32(flow)
    h, w, _ = flow.shape
    r = FitResult()
    lib().vo_fit_from_flow(_ptr(flow, C.c_float), h, w, int(step), MODES[requested_mode], C.byref(r))
    return np.array(list(r.matrix), np.float32).reshape(3, 3), MODE_NAMES[r.mode], float(r.confidence), float(r.residual)


def estimate_affine_partial2d(src, dst, thresh=2.0, max_iters=2000, confidence=0.992, refine_iters=10):
    src, dst = _f32(src).reshape(-1, 2), _f32(dst).reshape(-1, 2)
    n = src.shape[0]
    m = np.zeros(6, np.float64)
    inl = np.zeros(n, np.uint8)
    lib().vo_estimate_affine_partial2d.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_int, C.c_double, C.c_int, C.c_void_p, C.c_void_p]
    ok = lib().vo_estimate_affine_partial2d(src.ctypes.data, dst.ctypes.data, n, thresh, max_iters, confidence, refine_iters, m.ctypes.data, inl.ctypes.data)
    return (m.reshape(2, 3) if ok else None), inl


def find_homography(src, dst, thresh=2.5, max_iters=2000, confidence=0.992):
    src, dst = _f32(src).reshape(-1, 2), _f32(dst).reshape(-1, 2)
    n = src.shape[0]
    m = np.zeros(9, np.float64)
    inl = np.zeros(n, np.uint8)
    lib().vo_find_homography_ransac.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_int, C.c_double, C.c_void_p, C.c_void_p]
    ok = lib().vo_find_homography_ransac(src.ctypes.data, dst.ctypes.data, n, thresh, max_iters, confidence, m.ctypes.data, inl.ctypes.data)
    return (m.reshape(3, 3) if ok else None), inl


# ---------------------------------------------------------------- crop coverage analysis
def crop_analysis(matrices, src_size, out_size):
    m = _f32(matrices).reshape(-1, 9)
    n = m.shape[0]
    sw, sh = int(src_size[0]), int(src_size[1])
    ow, oh = int(out_size[0]), int(out_size[1])
    bbox = np.zeros((n, 4), np.int32)
    common = np.zeros((oh, ow), np.uint8)
    lib().vo_crop_analysis(_ptr(m, C.c_float), n, sh, sw, oh, ow, _ptr(bbox, C.c_int32), _ptr(common, C.c_uint8))
    return bbox, common


# ---------------------------------------------------------------- Classic estimator (GFTT + pyramidal LK)
GFTT = dict(max_corners=400, quality=0.01, min_distance=7.0, block=21)   # classic.py:76-83
LK = dict(win=31, max_level=3, max_count=50, epsilon=0.01)               # classic.py:88-96


def min_eigen_val(gray, block=21):
    gray = np.ascontiguousarray(gray, dtype=np.uint8)
    h, w = gray.shape
    eig = np.empty((h, w), np.float32)
    lib().vo_min_eigen_val(_ptr(gray, C.c_uint8), h, w, int(block), _ptr(eig, C.c_float))
    return eig


def good_features(gray, max_corners=400, quality=0.01, min_distance=7.0, block=21):
    """cv2.goodFeaturesToTrack(gray, ...) -> [n,2] f32 (x, y), strongest first."""
    gray = np.ascontiguousarray(gray, dtype=np.uint8)
    h, w = gray.shape
    cap = max_corners if max_corners > 0 else h * w
    pts = np.zeros((cap, 2), np.float32)
    fn = lib().vo_good_features
    fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, C.c_void_p]
    n = fn(gray.ctypes.data, h, w, int(max_corners), float(quality), float(min_distance), int(block), pts.ctypes.data)
    return pts[:n].copy()


def pyr_down(gray):
    gray = np.ascontiguousarray(gray, dtype=np.uint8)
    h, w = gray.shape
    out = np.empty(((h + 1) // 2, (w + 1) // 2), np.uint8)
    lib().vo_pyr_down_u8(_ptr(gray, C.c_uint8), h, w, _ptr(out, C.c_uint8))
    return out


def scharr_deriv(gray):
    gray = np.ascontiguousarray(gray, dtype=np.uint8)
    h, w = gray.shape
    out = np.empty((h, w, 2), np.int16)
    lib().vo_scharr_deriv(_ptr(gray, C.c_uint8), h, w, _ptr(out, C.c_int16))
    return out


def lk_levels(h, w, win=31, max_level=3):
    return int(lib().vo_lk_levels(int(h), int(w), int(win), int(max_level)))


def lk_track(prev, nxt, pts, win=31, max_level=3, max_count=50, epsilon=0.01):
    """cv2.calcOpticalFlowPyrLK(prev, nxt, pts, None, winSize=(win,win), maxLevel, criteria) -> (next [n,2], status [n])."""
    prev = np.ascontiguousarray(prev, dtype=np.uint8)
    nxt = np.ascontiguousarray(nxt, dtype=np.uint8)
    pts = _f32(pts).reshape(-1, 2)
    h, w = prev.shape
    n = pts.shape[0]
    out = np.zeros((n, 2), np.float32)
    status = np.zeros(n, np.uint8)
    fn = lib().vo_lk_track
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double,
                   C.c_void_p, C.c_void_p]
    fn(prev.ctypes.data, nxt.ctypes.data, h, w, pts.ctypes.data, n, int(win), int(max_level), int(max_count), float(epsilon),
       out.ctypes.data, status.ctypes.data)
    return out, status


def fit_all_modes_points(prev_pts, next_pts, status, requested_mode="similarity"):
    """Candidate fits of classic.py:105-160 on tracked points; same record layout as fit_all_modes."""
    prev_pts, next_pts = _f32(prev_pts).reshape(-1, 2), _f32(next_pts).reshape(-1, 2)
    status = np.ascontiguousarray(status, dtype=np.uint8)
    recs = (FitResult * 3)()
    nv = C.c_int()
    fn = lib().vo_fit_all_modes_points
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    fn(prev_pts.ctypes.data, next_pts.ctypes.data, status.ctypes.data, prev_pts.shape[0], MODES[requested_mode], recs, C.byref(nv))
    out = {}
    for mi in range(3):
        r = recs[mi]
        if r.mode < 0:
            continue
        out[MODE_NAMES[mi]] = {
            "matrix": np.array(list(r.matrix), np.float32).reshape(3, 3),
            "confidence": float(r.confidence),
            "residual": float(r.residual),
            "accepted": bool(r.valid),
        }
    return out, int(nv.value)


def optimal_dft_size(n):
    return int(lib().vo_optimal_dft_size(int(n)))


def phase_correlate_clip(gray, want_surface=False):
    """cv2.phaseCorrelate(gray[i].astype(f32), gray[i+1].astype(f32)) for every consecutive pair (flow.py:110-130)
    -> shifts f64 [N-1,3] = (tx, ty, response); with want_surface also the unshifted correlation surface of pair 0."""
    gray = np.ascontiguousarray(gray, dtype=np.uint8)
    n, h, w = gray.shape
    shifts = np.zeros((n - 1, 3), np.float64)
    surface = np.zeros((optimal_dft_size(h), optimal_dft_size(w)), np.float32) if want_surface else None
    fn = lib().vo_phase_correlate_clip
    fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    fn.restype = None
    fn(gray.ctypes.data, n, h, w, shifts.ctypes.data, surface.ctypes.data if want_surface else None)
    return (shifts, surface) if want_surface else shifts


def classic_estimate_pair(prev_gray, curr_gray, requested_mode="similarity"):
    """_estimate_motion_pair (classic.py:69-160): (matrix f32 3x3, used mode, confidence)."""
    feats = good_features(prev_gray, **GFTT)
    if feats.shape[0] < 12:
        return np.eye(3, dtype=np.float32), "translation", 0.0
    nxt, status = lk_track(prev_gray, curr_gray, feats, **LK)
    cands, _ = fit_all_modes_points(feats, nxt, status, requested_mode)
    order = {"perspective": ["perspective", "similarity", "translation"], "similarity": ["similarity", "translation"],
             "translation": ["translation"]}[requested_mode]
    for mode in order:
        c = cands.get(mode)
        if c is not None and c["accepted"]:
            return c["matrix"], mode, c["confidence"]
    return np.eye(3, dtype=np.float32), "translation", 0.0
