"""Real-OpenCV tier of the baseline / parity leg (TEST + MEASUREMENT INFRASTRUCTURE, like everything under oracle/).

SURVEY §8(d)(i) / BASELINE.md §2: if (and only if) `import cv2` works where the bench or the tests run, the reference's own
OpenCV calls -- same functions, same arguments -- are what the HIP path is timed beside (`cpu_baseline.kind = "opencv"`)
and compared with (`cv2_parity`).  No cv2 exists in the build container or on the GPU box today; nothing here installs or
fetches one.  The tier has the same method surface as `oracle.oracle` for the primitives on the C2 path, so
`bench.cpu_baseline` runs either provider through the same plan code.

Only `tests/` and `bench.py`'s cpu_baseline leg import this module; the product never does (tests/test_abi_cpu.py greps).

The reference call sites restated as calls (arguments are the contract; the code around them is this build's own):
    gray      nodes/stabilizer_utils.py:236-242,271-276   cvtColor(RGB2GRAY) f32 -> clip(*255) -> astype(u8) -> resize(INTER_AREA)
    DIS       nodes/video_stabilizer_flow.py:82-86,140     create(PRESET_MEDIUM); finest 2, patch 8, stride 4, spatial prop; calc(prev, curr, None)
    fit       nodes/video_stabilizer_flow.py:141-210       stride-8 grid; findHomography(RANSAC, 2.5, 2000, 0.992) >= 0.15;
                                                           estimateAffinePartial2D(RANSAC, 2.0, 2000, 0.992) >= 0.1; np.median
    warp      nodes/video_stabilizer_flow.py:561-588       warpPerspective(frame, M f32, size, INTER_LINEAR, BORDER_CONSTANT, rgb/255);
              nodes/motion_apply.py:94-115,173-190         warpPerspective(ones, M, size, INTER_NEAREST, BORDER_CONSTANT, 0) -> mask
"""

from __future__ import annotations

import importlib.util
import sys

import numpy as np

STANDIN_TAG = "oracle-standin"


def probe() -> dict:
    """What a bench line / test header states about the tier: {"cv2": "absent"} or version, threads, IPP."""
    mod = sys.modules.get("cv2")
    if mod is None:
        try:
            if importlib.util.find_spec("cv2") is None:
                return {"cv2": "absent"}
        except (ImportError, ValueError):
            return {"cv2": "absent"}
        try:
            import cv2 as mod  # noqa: WPS433
        except Exception as exc:   # a broken wheel is reported, not raised
            return {"cv2": "absent", "import_error": f"{type(exc).__name__}: {exc}"}
    version = str(getattr(mod, "__version__", "?"))
    out = {"cv2": version, "standin": version.startswith(STANDIN_TAG)}
    if hasattr(mod, "getNumThreads"):
        out["threads"] = int(mod.getNumThreads())
    ipp = getattr(mod, "ipp", None)
    if ipp is not None and hasattr(ipp, "useIPP"):
        out["ipp"] = bool(ipp.useIPP())
    return out


def available(allow_standin: bool = False) -> bool:
    info = probe()
    return info["cv2"] != "absent" and (allow_standin or not info.get("standin", False))


class Cv2Tier:
    """The C2 / C3 primitives answered by `cv2`, with oracle.oracle's method names and array conventions."""

    kind = "opencv"

    def __init__(self, threads: int | None = None):
        import cv2

        self.cv2 = cv2
        if threads is not None and hasattr(cv2, "setNumThreads"):
            cv2.setNumThreads(int(threads))

    # ---- F0 / F2
    def frame_max(self, frames):
        return np.array([float(f.max()) for f in frames], np.float32)             # utils.py:127: float(arr.max())

    def gray_for_estimation(self, frames, work_size):
        cv2 = self.cv2
        out = []
        for f in frames:
            g = cv2.cvtColor(np.ascontiguousarray(f, np.float32), cv2.COLOR_RGB2GRAY)
            g = np.clip(g * 255.0, 0, 255).astype(np.uint8)                       # utils.py:242 (truncates)
            if work_size is not None:
                g = cv2.resize(g, (int(work_size[0]), int(work_size[1])), interpolation=cv2.INTER_AREA)
            out.append(g)
        return np.stack(out)

    # ---- F3
    def make_dis(self):
        cv2 = self.cv2
        dis = cv2.DISOpticalFlow.create(cv2.DISOPTICAL_FLOW_PRESET_MEDIUM)
        dis.setFinestScale(2)
        dis.setPatchSize(8)
        dis.setPatchStride(4)
        dis.setUseSpatialPropagation(True)
        return dis

    def dis_flow_clip(self, gray):
        dis = self.make_dis()          # one object for the whole clip, as the reference keeps its backend (flow.py:315)
        return np.stack([dis.calc(gray[i], gray[i + 1], None) for i in range(len(gray) - 1)])

    # ---- F4 / F5: every candidate at or below the requested mode, in oracle.fit_all_modes' record format
    def fit_all_modes(self, flow, step=8, requested_mode="similarity"):
        cv2 = self.cv2
        h, w = flow.shape[:2]
        ys, xs = np.mgrid[0:h:step, 0:w:step]
        prev = np.stack([xs.ravel(), ys.ravel()], axis=1).astype(np.float32)
        curr = prev + flow[ys, xs].reshape(-1, 2)
        ok = np.isfinite(curr).all(axis=1)
        p, q = prev[ok], curr[ok]
        out = {}
        if len(p) < 12:
            return out, int(len(p)), int(len(prev))
        level = ("translation", "similarity", "perspective").index(requested_mode)
        if level >= 2 and len(p) >= 4:
            rec = {"matrix": np.eye(3, dtype=np.float32), "confidence": 0.0, "residual": 0.0, "accepted": False}
            H, inl = cv2.findHomography(p, q, method=cv2.RANSAC, ransacReprojThreshold=2.5, maxIters=2000, confidence=0.992)
            if H is not None and inl is not None:
                rec["confidence"] = float(inl.sum()) / float(len(p))
                if rec["confidence"] >= 0.15:
                    rec["residual"] = float(np.abs((p @ H[:2, :2].T + H[:2, 2]) - q).mean())     # affine part only (flow.py:174)
                    rec["matrix"], rec["accepted"] = H.astype(np.float32), True
            out["perspective"] = rec
        if level >= 1 and len(p) >= 3:
            rec = {"matrix": np.eye(3, dtype=np.float32), "confidence": 0.0, "residual": 0.0, "accepted": False}
            M, inl = cv2.estimateAffinePartial2D(p, q, method=cv2.RANSAC, ransacReprojThreshold=2.0, maxIters=2000, confidence=0.992)
            if M is not None:
                rec["confidence"] = float(inl.sum()) / float(len(p)) if inl is not None else 0.0
                if rec["confidence"] >= 0.1:
                    rec["residual"] = float(np.abs((p @ M[:, :2].T + M[:, 2]) - q).mean())
                    rec["matrix"], rec["accepted"] = np.vstack([M, [0.0, 0.0, 1.0]]).astype(np.float32), True
            out["similarity"] = rec
        d = np.median(q - p, axis=0).astype(np.float32)
        m = np.eye(3, dtype=np.float32)
        m[0, 2], m[1, 2] = d[0], d[1]
        out["translation"] = {"matrix": m, "confidence": float(len(p)) / float(len(prev)),
                              "residual": float(np.abs((p + d) - q).mean()), "accepted": True}
        return out, int(len(p)), int(len(prev))

    # ---- F13 / A3
    def _flag(self, interp):
        return self.cv2.INTER_CUBIC if interp == "bicubic" else self.cv2.INTER_LINEAR

    def warp_clip(self, src, matrices, out_size, interp="bilinear", border=(0.0, 0.0, 0.0), subpix=None, want_mask=True):
        cv2 = self.cv2
        n, sh, sw, _ = src.shape
        size = (int(out_size[0]), int(out_size[1]))
        bv = [float(v) for v in np.asarray(border, np.float32)]
        dst = np.empty((n, size[1], size[0], 3), np.float32)
        mask = np.empty((n, size[1], size[0]), np.float32) if want_mask else None
        counts = np.zeros(n, np.uint32)
        ones = np.ones((sh, sw), np.float32)
        for i in range(n):
            m = np.asarray(matrices[i], np.float32).reshape(3, 3)
            dst[i] = cv2.warpPerspective(np.ascontiguousarray(src[i], np.float32), m, size, flags=self._flag(interp),
                                         borderMode=cv2.BORDER_CONSTANT, borderValue=bv)
            if want_mask:
                cov = cv2.warpPerspective(ones, m, size, flags=cv2.INTER_NEAREST, borderMode=cv2.BORDER_CONSTANT, borderValue=0.0)
                mk = 1.0 - (cov > 0.5).astype(np.float32)
                mk[mk < 1e-3] = 0.0
                mask[i] = mk
                counts[i] = int(np.count_nonzero(mk))
        return dst, mask, counts

    # ---- A5 (motion_apply.py:137-202): S full warps accumulated in f32, divided by S
    def warp_blur_clip(self, src, matrices64, out_size, blur, samples, interp="bilinear", border=(0.0, 0.0, 0.0), subpix=None,
                       want_mask=True):
        cv2 = self.cv2
        n, sh, sw, _ = src.shape
        size = (int(out_size[0]), int(out_size[1]))
        bv = [float(v) for v in np.asarray(border, np.float32)]
        m64 = np.asarray(matrices64, np.float64).reshape(n, 3, 3)
        ts = np.linspace(0.0, float(blur), int(samples))
        dst = np.empty((n, size[1], size[0], 3), np.float32)
        mask = np.empty((n, size[1], size[0]), np.float32) if want_mask else None
        ones = np.ones((sh, sw), np.float32)
        for i in range(n):
            if n == 1:
                mats = [m64[0]]
            else:
                delta = (m64[i + 1] - m64[i]) if i + 1 < n else (m64[i] - m64[i - 1])
                mats = [m64[i] + delta * t for t in ts]
            acc = np.zeros((size[1], size[0], 3), np.float32)
            cov = np.zeros((size[1], size[0]), np.float32)
            for mk in mats:
                m = mk.astype(np.float32)
                acc += cv2.warpPerspective(np.ascontiguousarray(src[i], np.float32), m, size, flags=self._flag(interp),
                                           borderMode=cv2.BORDER_CONSTANT, borderValue=bv)
                if want_mask:
                    c = cv2.warpPerspective(ones, m, size, flags=cv2.INTER_NEAREST, borderMode=cv2.BORDER_CONSTANT, borderValue=0.0)
                    cov += (c > 0.5).astype(np.float32)
            dst[i] = acc / float(samples)
            if want_mask:
                mk = 1.0 - cov / float(samples)
                mk[mk < 1e-3] = 0.0
                mask[i] = mk
        return dst, mask


def _stats(a, b):
    d = np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64))
    return {"max": float(d.max()), "mean": float(d.mean()), "frac_over_1e-3": float((d > 1e-3).mean())}


def primitive_parity(vo, tier, frames, work_size, border, pairs: int = 4) -> dict:
    """Oracle restatement vs OpenCV, primitive by primitive, on the first `pairs`+1 frames of a clip [n,H,W,3] f32: the
    first thing to run when a real cv2 is reachable (DESIGN §5's two named risks show here as `gray_u8_differing` and as
    which `warp_*_subpix` row is exact).  Returns counts / error statistics; asserts nothing."""
    n = min(len(frames), pairs + 1)
    fr = np.ascontiguousarray(frames[:n], np.float32)
    h, w = fr.shape[1:3]
    out = {"frames": n, "size": [w, h]}
    g_cv = tier.gray_for_estimation(fr, work_size)
    g_vo = vo.gray_for_estimation(fr, work_size)
    out["gray_u8_differing"] = int(np.count_nonzero(g_cv != g_vo))
    out["gray_u8_max_abs"] = int(np.abs(g_cv.astype(np.int16) - g_vo.astype(np.int16)).max())
    # DIS on OpenCV's own gray (isolates DIS from a gray difference)
    f_cv = tier.dis_flow_clip(g_cv)
    f_vo = vo.dis_flow_clip(g_cv)
    epe = np.hypot(f_cv[..., 0] - f_vo[..., 0], f_cv[..., 1] - f_vo[..., 1])
    out["flow_epe_px"] = {"max": float(epe.max()), "mean": float(epe.mean()), "p99": float(np.percentile(epe, 99)),
                          "frac_over_1e-3": float((epe > 1e-3).mean())}
    grid = epe[:, ::8, ::8]
    out["flow_epe_px_at_stride8_samples"] = {"max": float(grid.max()), "mean": float(grid.mean())}
    # fits on OpenCV's own flow
    for mode in ("similarity", "perspective"):
        deltas, confs = [], []
        for i in range(n - 1):
            a = tier.fit_all_modes(f_cv[i], 8, mode)[0].get(mode)
            b = vo.fit_all_modes(f_cv[i], 8, mode)[0].get(mode)
            if a is None or b is None or not (a["accepted"] and b["accepted"]):
                deltas.append(float("nan"))
                continue
            deltas.append(float(np.abs(a["matrix"].astype(np.float64) - b["matrix"]).max()))
            confs.append(abs(a["confidence"] - b["confidence"]))
        out[f"fit_{mode}_matrix_max_abs"] = deltas
        out[f"fit_{mode}_confidence_max_abs"] = float(max(confs)) if confs else None
    # warp
    m = np.array([[1.01 * np.cos(0.02), -1.01 * np.sin(0.02), 3.37], [1.01 * np.sin(0.02), 1.01 * np.cos(0.02), -2.61], [0, 0, 1.0]], np.float32)
    mats = np.stack([m] * 1)
    for interp in ("bilinear", "bicubic"):
        d_cv, k_cv, _ = tier.warp_clip(fr[:1], mats, (w, h), interp=interp, border=border)
        for subpix in (("q5", "exact") if interp == "bilinear" else ("q5",)):
            d_vo, k_vo, _ = vo.warp_clip(fr[:1], mats, (w, h), interp=interp, border=border, subpix=subpix)
            out[f"warp_{interp}_{subpix}"] = _stats(d_cv, d_vo)
        out[f"mask_{interp}_differing"] = int(np.count_nonzero(k_cv != k_vo))
    return out
