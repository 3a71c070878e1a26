"""A `cv2` module for the build container, served by the CPU oracle (TEST INFRASTRUCTURE).

Purpose: OpenCV is absent from this image, so the reference's pipelines (`_stabilize_frames`, `apply_motion`, the crop
solver) stop at their first `cv2.*` call.  With this module in `sys.modules["cv2"]` the reference's OWN Python -- its
control flow, NumPy arithmetic, sticky-mode walk, trajectory, framing, crop bisection, meta assembly, progress ticks --
runs end to end here, with each `cv2` primitive answered by `oracle/` (the C restatement of the published OpenCV
algorithm).  `tests/golden/make_e2e_golden.py` uses it to write fixtures; nothing else imports it.

What this does and does not pin: the fixtures pin everything BETWEEN the OpenCV calls against the reference's own code
(flow.py:213-640, motion_apply.py:297-429, stabilizer_utils.py:448-837).  They do NOT pin the OpenCV primitives: those
are still the oracle's from-memory restatement ("parity unpinned" in DESIGN.md section 5).

Only the functions, flags and argument forms the reference actually uses are provided; anything else raises.
"""

from __future__ import annotations

import sys
import types

import numpy as np

from oracle import oracle as vo

# ---- constants (values as in OpenCV 4.x headers) -------------------------------------------------------------------
INTER_NEAREST, INTER_LINEAR, INTER_CUBIC, INTER_AREA = 0, 1, 2, 3
BORDER_CONSTANT = 0
COLOR_RGB2GRAY = 7
CV_64F = 6
MORPH_RECT = 0
RANSAC = 8
DISOPTICAL_FLOW_PRESET_ULTRAFAST, DISOPTICAL_FLOW_PRESET_FAST, DISOPTICAL_FLOW_PRESET_MEDIUM = 0, 1, 2
__version__ = "oracle-standin (not OpenCV)"

CALLS: dict = {}   # name -> number of calls, so the generator can state which primitives a scenario exercised


def _count(name: str) -> None:
    CALLS[name] = CALLS.get(name, 0) + 1


# ---- color / resize -------------------------------------------------------------------------------------------------
def cvtColor(src, code):
    if code != COLOR_RGB2GRAY:
        raise NotImplementedError(f"cv2 stand-in: cvtColor code {code}")
    src = np.asarray(src)
    if src.dtype != np.float32 or src.ndim != 3 or src.shape[2] != 3:
        raise NotImplementedError(f"cv2 stand-in: cvtColor on {src.dtype} {src.shape}")
    _count("cvtColor")
    return vo.rgb2gray_f32(src)


def resize(src, dsize, interpolation=INTER_LINEAR):
    src = np.asarray(src)
    if interpolation != INTER_AREA or src.dtype != np.uint8 or src.ndim != 2:
        raise NotImplementedError("cv2 stand-in: resize is provided for uint8 single-channel INTER_AREA only")
    _count("resize")
    return vo.resize_area_u8(src, dsize)


# ---- warpPerspective ------------------------------------------------------------------------------------------------
def warpPerspective(src, M, dsize, flags=INTER_LINEAR, borderMode=BORDER_CONSTANT, borderValue=0.0):
    if borderMode != BORDER_CONSTANT:
        raise NotImplementedError("cv2 stand-in: BORDER_CONSTANT only")
    src = np.asarray(src)
    # OpenCV: M.convertTo(matM, CV_64F); invert(matM, matM) -- whatever dtype the caller passes
    inverse = vo.invert3x3(np.asarray(M, dtype=np.float64))
    if flags == INTER_NEAREST:
        # the reference only ever warps an all-ones float32 plane with border 0 to obtain coverage
        if src.ndim != 2 or src.dtype != np.float32 or not bool((src == 1.0).all()) or float(np.max(np.abs(np.atleast_1d(borderValue)))) != 0.0:
            raise NotImplementedError("cv2 stand-in: INTER_NEAREST is provided for the all-ones coverage plane only")
        _count("warpPerspective.nearest")
        dummy = np.zeros(src.shape + (3,), np.float32)
        _, cov = vo.warp_frame_inv(dummy, inverse, dsize, "bilinear", (0.0, 0.0, 0.0))
        return cov
    if flags not in (INTER_LINEAR, INTER_CUBIC):
        raise NotImplementedError(f"cv2 stand-in: warpPerspective flags {flags}")
    if src.ndim != 3 or src.shape[2] != 3 or src.dtype != np.float32:
        raise NotImplementedError(f"cv2 stand-in: warpPerspective on {src.dtype} {src.shape}")
    border = np.asarray(borderValue, dtype=np.float64).reshape(-1)
    if border.size == 1:
        border = np.repeat(border, 3)
    # cv::Scalar is 4 doubles; the kernels convert the border to the image type
    border = border[:3].astype(np.float32)
    _count("warpPerspective.cubic" if flags == INTER_CUBIC else "warpPerspective.linear")
    dst, _ = vo.warp_frame_inv(src, inverse, dsize, "bicubic" if flags == INTER_CUBIC else "bilinear", border)
    return dst


# ---- morphology / integral (exact integer / min-max operations, plain NumPy) ----------------------------------------
def getStructuringElement(shape, ksize):
    if shape != MORPH_RECT:
        raise NotImplementedError("cv2 stand-in: MORPH_RECT only")
    return np.ones((int(ksize[1]), int(ksize[0])), np.uint8)


def _morph(src, kernel, iterations, op):
    src = np.asarray(src)
    kernel = np.asarray(kernel)
    if src.ndim != 2 or kernel.shape[0] % 2 != 1 or kernel.shape[1] % 2 != 1 or not bool((kernel != 0).all()):
        raise NotImplementedError("cv2 stand-in: morphology with a full odd rectangular kernel on a single plane only")
    ry, rx = kernel.shape[0] // 2, kernel.shape[1] // 2
    out = src
    for _ in range(int(iterations)):
        # default border of cv2.erode / cv2.dilate: pixels outside the image never win (erode: +inf, dilate: -inf)
        if op == "erode":
            fill = np.array(np.inf if src.dtype.kind == "f" else np.iinfo(src.dtype).max, dtype=src.dtype)
        else:
            fill = np.array(-np.inf if src.dtype.kind == "f" else np.iinfo(src.dtype).min, dtype=src.dtype)
        padded = np.pad(out, ((ry, ry), (rx, rx)), mode="constant", constant_values=fill)
        acc = None
        for dy in range(kernel.shape[0]):
            for dx in range(kernel.shape[1]):
                win = padded[dy:dy + out.shape[0], dx:dx + out.shape[1]]
                acc = win if acc is None else (np.minimum(acc, win) if op == "erode" else np.maximum(acc, win))
        out = np.ascontiguousarray(acc)
    return out


def erode(src, kernel, iterations=1):
    _count("erode")
    return _morph(src, kernel, iterations, "erode")


def dilate(src, kernel, iterations=1):
    _count("dilate")
    return _morph(src, kernel, iterations, "dilate")


def integral(src, sdepth=-1):
    src = np.asarray(src)
    if sdepth != CV_64F or src.ndim != 2:
        raise NotImplementedError("cv2 stand-in: integral(sdepth=CV_64F) on a single plane only")
    _count("integral")
    out = np.zeros((src.shape[0] + 1, src.shape[1] + 1), np.float64)
    out[1:, 1:] = np.cumsum(np.cumsum(src.astype(np.float64), axis=0), axis=1)   # exact: small integers
    return out


# ---- DIS optical flow -----------------------------------------------------------------------------------------------
class _DIS:
    """cv2.DISOpticalFlow object: PRESET_MEDIUM defaults, the four setters the reference calls (flow.py:82-86), and a
    calc() that keeps the auto-selected finest scale in the object exactly as OpenCV does for tiny images."""

    def __init__(self, preset):
        if preset != DISOPTICAL_FLOW_PRESET_MEDIUM:
            raise NotImplementedError("cv2 stand-in: DIS PRESET_MEDIUM only")
        # MEDIUM: finest 1, patch 8, stride 3, 25 GD iterations, 5 VR iterations (the setters below overwrite three)
        self.params = vo.dis_params(finest_scale=1, patch_stride=3)

    def setFinestScale(self, v):
        self.params.finest_scale = int(v)

    def setPatchSize(self, v):
        self.params.patch_size = int(v)

    def setPatchStride(self, v):
        self.params.patch_stride = int(v)

    def setUseSpatialPropagation(self, v):
        self.params.use_spatial_prop = 1 if v else 0

    def calc(self, i0, i1, flow):
        if flow is not None:
            raise NotImplementedError("cv2 stand-in: DIS initial flow is not supported")
        _count("DISOpticalFlow.calc")
        return vo.dis_flow_stateful(i0, i1, self.params)


DIS_DISABLED = None   # a message: DISOpticalFlow.create raises it (drives the reference's flow.py:90-107 fallback selection)


class DISOpticalFlow:
    @staticmethod
    def create(preset=DISOPTICAL_FLOW_PRESET_FAST):
        if DIS_DISABLED:
            raise RuntimeError(DIS_DISABLED)
        return _DIS(preset)


# ---- robust fits ----------------------------------------------------------------------------------------------------
def findHomography(srcPoints, dstPoints, method=0, ransacReprojThreshold=3.0, maxIters=2000, confidence=0.995):
    if method != RANSAC:
        raise NotImplementedError("cv2 stand-in: findHomography(RANSAC) only")
    _count("findHomography")
    H, inl = vo.find_homography(srcPoints, dstPoints, ransacReprojThreshold, maxIters, confidence)
    if H is None:
        return None, None
    return H, inl.reshape(-1, 1)


def estimateAffinePartial2D(from_, to, method=RANSAC, ransacReprojThreshold=3.0, maxIters=2000, confidence=0.99, refineIters=10):
    if method != RANSAC:
        raise NotImplementedError("cv2 stand-in: estimateAffinePartial2D(RANSAC) only")
    _count("estimateAffinePartial2D")
    M, inl = vo.estimate_affine_partial2d(from_, to, ransacReprojThreshold, maxIters, confidence, refineIters)
    if M is None:
        return None, inl.reshape(-1, 1)
    return M, inl.reshape(-1, 1)


def phaseCorrelate(a, b):
    a, b = np.asarray(a), np.asarray(b)
    _count("phaseCorrelate")
    gray = np.stack([a, b]).astype(np.uint8)
    if not (np.array_equal(gray[0].astype(np.float32), a) and np.array_equal(gray[1].astype(np.float32), b)):
        raise NotImplementedError("cv2 stand-in: phaseCorrelate on float images holding uint8 values only")
    tx, ty, resp = vo.phase_correlate_clip(gray)[0]
    return (float(tx), float(ty)), float(resp)


# ---- Classic estimator primitives (nodes/video_stabilizer_classic.py:76-96) ------------------------------------------
TERM_CRITERIA_COUNT, TERM_CRITERIA_EPS = 1, 2


def goodFeaturesToTrack(image, maxCorners, qualityLevel, minDistance, mask=None, blockSize=3, **kw):
    if mask is not None or kw:
        raise NotImplementedError("cv2 stand-in: goodFeaturesToTrack without mask / extra options only")
    _count("goodFeaturesToTrack")
    pts = vo.good_features(image, maxCorners, qualityLevel, minDistance, blockSize)
    return pts.reshape(-1, 1, 2) if len(pts) else None


def calcOpticalFlowPyrLK(prevImg, nextImg, prevPts, nextPts, winSize=(21, 21), maxLevel=3, criteria=(3, 30, 0.01), **kw):
    if nextPts is not None or kw or winSize[0] != winSize[1]:
        raise NotImplementedError("cv2 stand-in: calcOpticalFlowPyrLK without initial points / flags only")
    _count("calcOpticalFlowPyrLK")
    pts = np.asarray(prevPts, np.float32).reshape(-1, 2)
    out, status = vo.lk_track(prevImg, nextImg, pts, winSize[0], maxLevel, criteria[1], criteria[2])
    return out.reshape(-1, 1, 2), status.reshape(-1, 1), None


# ---- drawing / synthesis helpers used ONLY by the reference's check scripts to make their test clips ----------------
# (plain NumPy, no claim of matching OpenCV's rasteriser bit for bit: they produce inputs, not results under test)
BORDER_REFLECT = 2


def rectangle(img, pt1, pt2, color, thickness=-1):
    if thickness != -1:
        raise NotImplementedError("cv2 stand-in: filled rectangles only")
    x0, y0 = pt1
    x1, y1 = pt2
    img[max(y0, 0):y1 + 1, max(x0, 0):x1 + 1] = np.asarray(color, img.dtype)[: img.shape[2]]
    return img


def circle(img, center, radius, color, thickness=-1):
    if thickness != -1:
        raise NotImplementedError("cv2 stand-in: filled circles only")
    yy, xx = np.mgrid[0:img.shape[0], 0:img.shape[1]]
    inside = (xx - center[0]) ** 2 + (yy - center[1]) ** 2 <= radius * radius
    img[inside] = np.asarray(color, img.dtype)[: img.shape[2]]
    return img


def getRotationMatrix2D(center, angle, scale):
    a = np.deg2rad(angle)
    alpha, beta = scale * np.cos(a), scale * np.sin(a)
    return np.array([[alpha, beta, (1 - alpha) * center[0] - beta * center[1]],
                     [-beta, alpha, beta * center[0] + (1 - alpha) * center[1]]], np.float64)


def warpAffine(src, M, dsize, flags=INTER_LINEAR, borderMode=BORDER_CONSTANT, borderValue=0):
    if flags != INTER_LINEAR or borderMode != BORDER_REFLECT:
        raise NotImplementedError("cv2 stand-in: warpAffine(INTER_LINEAR, BORDER_REFLECT) only (clip synthesis of the check scripts)")
    src = np.asarray(src, np.float32)
    h, w = src.shape[:2]
    full = np.vstack([np.asarray(M, np.float64), [0.0, 0.0, 1.0]])
    inv = np.linalg.inv(full)
    yy, xx = np.mgrid[0:dsize[1], 0:dsize[0]].astype(np.float64)
    sx = inv[0, 0] * xx + inv[0, 1] * yy + inv[0, 2]
    sy = inv[1, 0] * xx + inv[1, 1] * yy + inv[1, 2]
    x0, y0 = np.floor(sx).astype(np.int64), np.floor(sy).astype(np.int64)
    fx, fy = (sx - x0)[..., None].astype(np.float32), (sy - y0)[..., None].astype(np.float32)

    def reflect(i, n):   # BORDER_REFLECT: fedcba|abcdefgh|hgfedcb
        i = np.mod(i, 2 * n)
        return np.where(i >= n, 2 * n - 1 - i, i)

    xa, xb, ya, yb = reflect(x0, w), reflect(x0 + 1, w), reflect(y0, h), reflect(y0 + 1, h)
    out = (src[ya, xa] * (1 - fx) * (1 - fy) + src[ya, xb] * fx * (1 - fy) + src[yb, xa] * (1 - fx) * fy + src[yb, xb] * fx * fy)
    return out.astype(np.float32)


def install() -> types.ModuleType:
    """Register this module as `cv2` (and make `import cv2.optflow` fail the way a stock opencv-python-headless does)."""
    mod = sys.modules[__name__]
    sys.modules["cv2"] = mod
    sys.modules.pop("cv2.optflow", None)
    return mod
