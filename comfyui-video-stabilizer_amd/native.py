"""ctypes binding of libvstab.so (include/vstab.h) -- the only compute path of this package.

There is no CPU fallback: if the HIP library is missing or no MI355X is visible every
entry point raises.  torch tensors are used as the device-memory container only.
"""

from __future__ import annotations

import ctypes as C
import os
from pathlib import Path
from typing import Any, Dict, Optional

import numpy as np

_PKG_DIR = Path(__file__).resolve().parent
# VSTAB_LIB: another build of the library (A/B measurements of kernel variants inside one GPU session, tools/*.py)
LIB_PATH = Path(os.environ["VSTAB_LIB"]).resolve() if os.environ.get("VSTAB_LIB") else _PKG_DIR / "lib" / "libvstab.so"

INTERP = {"bilinear": 0, "bicubic": 1}
SUBPIX = {"q5": 0, "exact": 1}
MODES = {"translation": 0, "similarity": 1, "perspective": 2}
MODE_NAMES = ("translation", "similarity", "perspective")

# Sub-pixel model used by the node path; see include/vstab.h (vstab_subpix) and DESIGN.md.
DEFAULT_SUBPIX = os.environ.get("VSTAB_SUBPIX", "q5")


class VstabError(RuntimeError):
    pass


class FitRecord(C.Structure):
    _fields_ = [
        ("matrix", C.c_float * 9),
        ("confidence", C.c_double),
        ("residual", C.c_double),
        ("accepted", C.c_int32),
        ("computed", C.c_int32),
        ("valid_points", C.c_int32),
        ("total_points", C.c_int32),
    ]


# numpy view of vstab_fit_record (include/vstab.h): same layout as the ctypes structure above
FIT_DTYPE = np.dtype({
    "names": ["matrix", "confidence", "residual", "accepted", "computed", "valid_points", "total_points"],
    "formats": [(np.float32, (9,)), np.float64, np.float64, np.int32, np.int32, np.int32, np.int32],
    "offsets": [0, 40, 48, 56, 60, 64, 68],
    "itemsize": 72,
})
assert C.sizeof(FitRecord) == FIT_DTYPE.itemsize


def fit_table_from_dicts(records) -> np.ndarray:
    """List of {mode_name: candidate dict} (the oracle-side shape used in tests) -> structured [P,3] table."""
    table = np.zeros((len(records), 3), FIT_DTYPE)
    table["matrix"][:] = np.eye(3, dtype=np.float32).reshape(9)
    for p, entry in enumerate(records):
        for mi, name in enumerate(MODE_NAMES):
            cand = entry.get(name)
            if cand is None:
                continue
            row = table[p, mi]
            row["matrix"] = np.asarray(cand["matrix"], np.float32).reshape(9)
            row["confidence"] = cand["confidence"]
            row["residual"] = cand["residual"]
            row["accepted"] = 1 if cand["accepted"] else 0
            row["computed"] = 1
            row["valid_points"] = cand.get("valid_points", 0)
            row["total_points"] = cand.get("total_points", 0)
    return table


def fit_table_to_dicts(table: np.ndarray):
    out = []
    for p in range(table.shape[0]):
        entry = {}
        for mi, name in enumerate(MODE_NAMES):
            r = table[p, mi]
            if not r["computed"]:
                continue
            entry[name] = {
                "matrix": np.array(r["matrix"], dtype=np.float32).reshape(3, 3),
                "confidence": float(r["confidence"]),
                "residual": float(r["residual"]),
                "accepted": bool(r["accepted"]),
                "valid_points": int(r["valid_points"]),
                "total_points": int(r["total_points"]),
            }
        out.append(entry)
    return out


_SIGNATURES = {
    "vstab_abi_version": (C.c_int, []),
    "vstab_last_pad_counts": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "vstab_flow_plan_zero_counts": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    "vstab_last_frame_peaks": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "vstab_test_hooks": (C.c_int, []),
    "vstab_last_error": (C.c_char_p, []),
    "vstab_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int]),
    "vstab_destroy": (C.c_int, [C.c_void_p]),
    "vstab_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "vstab_synchronize": (C.c_int, [C.c_void_p]),
    "vstab_set_timing": (C.c_int, [C.c_void_p, C.c_int]),
    "vstab_last_kernel_ms": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_float)]),
    "vstab_kernel_ms_stats": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_int)]),
    "vstab_upload": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "vstab_download": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "vstab_upload_f32_coded": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "vstab_upload_u8_as_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "vstab_download_mask_levels": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.POINTER(C.c_int)]),
    "vstab_download_mask_coded": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_int)]),
    "vstab_warp_batch": (
        C.c_int,
        [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p,
         C.c_int, C.c_void_p, C.c_void_p, C.c_void_p],
    ),
    "vstab_warp_blur_batch": (
        C.c_int,
        [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
         C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p],
    ),
    "vstab_warp_blur_clip_batch": (
        C.c_int,
        [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int,
         C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p],
    ),
    "vstab_blur_sample_matrices": (
        C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "vstab_common_coverage": (
        C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "vstab_gray_downscale": (
        C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "vstab_gray_downscale_range": (
        C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "vstab_frame_range": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "vstab_apply_value_range": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "vstab_dis_flow_batch": (
        C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int]),
    "vstab_dis_set_clip_start": (C.c_int, [C.c_void_p, C.c_int]),
    "vstab_sample_fit_batch": (
        C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "vstab_sample_fit_batch_begin": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "vstab_fit_records_device": (C.c_void_p, [C.c_void_p]),
    "vstab_sample_fit_batch_end": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "vstab_flow_plan_device": (
        C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_int,
                  C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int]),
    "vstab_fit_records_copy": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    "vstab_flow_plan_result": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "vstab_warp_batch_planned": (
        C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int,
                  C.c_void_p, C.c_void_p, C.c_void_p]),
    "vstab_gftt_batch": (
        C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, C.c_void_p,
                  C.c_void_p]),
    "vstab_lk_levels": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "vstab_lk_track_batch": (
        C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                  C.c_int, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]),
    "vstab_points_fit_batch": (
        C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "vstab_phase_correlate_batch": (
        C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "vstab_host_math": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "vstab_transitions_to_params": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "vstab_params_to_matrices": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "vstab_bounding_boxes": (C.c_int, [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_void_p, C.c_void_p]),
    "vstab_crop_analysis": (
        C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "vstab_trajectory": (
        C.c_int,
        [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_int, C.c_void_p,
         C.c_void_p],
    ),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)

_lib = None


def load_library():
    """Load libvstab.so; raises VstabError if it has not been built (no fallback exists)."""
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise VstabError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). This package has no CPU fallback."
            )
        # torch first: its wheel bundles its own libamdhip64.so.7 (+ libhsa-runtime64), and libvstab.so needs the same
        # SONAME.  Loaded after torch, libvstab binds to torch's copy: one HIP runtime in the process, torch's streams
        # and allocations are ours.  Loaded BEFORE torch it would pull in /opt/rocm's copy, torch would then find the
        # SONAME taken and run on a runtime that does not match its other bundled libraries ("no ROCm-capable device").
        import torch  # noqa: F401

        lib = C.CDLL(str(LIB_PATH))
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError here means the ABI and the header diverged
            fn.restype = res
            fn.argtypes = args
        if lib.vstab_abi_version() != 1:
            raise VstabError(f"libvstab ABI version {lib.vstab_abi_version()} != 1")
        _lib = lib
    return _lib


HOST_OPS = {"sqrt": 0, "atan2": 1, "log": 2, "exp": 3, "cos": 4, "sin": 5}


def host_math(op: str, a, b=None) -> np.ndarray:
    """Element-wise libm call over fp64 arrays (vstab_host_math): the functions Python's math module binds."""
    a = np.ascontiguousarray(a, dtype=np.float64)
    out = np.empty_like(a)
    bb = np.ascontiguousarray(b, dtype=np.float64) if b is not None else None
    rc = load_library().vstab_host_math(HOST_OPS[op], a.ctypes.data, bb.ctypes.data if bb is not None else None, a.size, out.ctypes.data)
    _check(rc, "vstab_host_math")
    return out


PARAM_COUNT = {"translation": 2, "similarity": 4, "perspective": 8}


def transitions_to_params(work_mats, base_mode: str, source_size=None, working_size=None):
    """flow.py:340-346 for a clip (vstab_transitions_to_params): f32 [P,3,3] at working resolution ->
    (f32 [P,3,3] at full resolution, f64 [P,K] parameter deltas).  working_size None: estimated at full size."""
    m = np.ascontiguousarray(work_mats, dtype=np.float32).reshape(-1, 9)
    count = m.shape[0]
    full = np.empty((count, 9), np.float32)
    params = np.empty((count, PARAM_COUNT[base_mode]), np.float64)
    if working_size is not None:
        sx = working_size[0] / float(source_size[0])
        sy = working_size[1] / float(source_size[1])
        up = np.array([1.0 / sx, 1.0 / sy, 1.0], np.float64)
        down = np.array([sx, sy, 1.0], np.float64)
        up_p, down_p = up.ctypes.data, down.ctypes.data
    else:
        up_p = down_p = None
    _check(load_library().vstab_transitions_to_params(m.ctypes.data, count, MODES[base_mode], up_p, down_p, full.ctypes.data,
                                                      params.ctypes.data), "vstab_transitions_to_params")
    return full.reshape(count, 3, 3), params


def params_to_matrices(params, base_mode: str) -> np.ndarray:
    """_params_to_matrix for a clip (vstab_params_to_matrices): f64 [N,K] -> f32 [N,3,3]."""
    p = np.ascontiguousarray(params, dtype=np.float64).reshape(-1, PARAM_COUNT[base_mode])
    out = np.empty((p.shape[0], 3, 3), np.float32)
    _check(load_library().vstab_params_to_matrices(p.ctypes.data, p.shape[0], MODES[base_mode], out.ctypes.data),
           "vstab_params_to_matrices")
    return out


def bounding_boxes(matrices, width: int, height: int):
    """_compute_bounding_boxes for a stacked f32 [N,3,3] array (vstab_bounding_boxes) -> (mins [N,2], maxs [N,2]) f64."""
    m = np.ascontiguousarray(matrices, dtype=np.float32).reshape(-1, 9)
    mins = np.empty((m.shape[0], 2), np.float64)
    maxs = np.empty((m.shape[0], 2), np.float64)
    _check(load_library().vstab_bounding_boxes(m.ctypes.data, m.shape[0], float(width), float(height), mins.ctypes.data,
                                               maxs.ctypes.data), "vstab_bounding_boxes")
    return mins, maxs


def blur_sample_matrices(matrices64, first: int, count: int, blur: float, samples: int) -> np.ndarray:
    """The float32 shutter-sample matrices the blur warp uses for frames [first, first+count) of a clip
    (vstab_blur_sample_matrices; motion_apply.py:125-134 + the cast of :172) -> [count, S', 3, 3] float32."""
    m = np.ascontiguousarray(matrices64, dtype=np.float64).reshape(-1, 9)
    ts = np.ascontiguousarray(np.linspace(0.0, float(blur), int(samples), dtype=np.float64))
    per_frame = 1 if m.shape[0] <= 1 else int(samples)
    out = np.zeros((int(count), per_frame, 3, 3), np.float32)
    rc = load_library().vstab_blur_sample_matrices(m.ctypes.data, m.shape[0], int(first), int(count), ts.ctypes.data, int(samples),
                                                   out.ctypes.data)
    _check(rc, "vstab_blur_sample_matrices")
    return out


def _check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load_library().vstab_last_error()
        raise VstabError(f"{what} failed ({rc}): {msg.decode('utf-8', 'replace') if msg else 'unknown error'}")


def _coded_transfers():
    return os.environ.get("VSTAB_XFER_CODED", "1") not in ("0", "false", "False")


def _dev_ptr(t) -> int:
    return int(t.data_ptr())


class Context:
    """One libvstab context bound to one GPU (one process per GPU in multi-GPU runs)."""

    def __init__(self, device: Optional[int] = None):
        import torch

        if not torch.cuda.is_available():
            raise VstabError("no MI355X / HIP device visible: the stabilizer hot path has no CPU fallback")
        self.torch = torch
        self.device_index = torch.cuda.current_device() if device is None else int(device)
        self.device = torch.device("cuda", self.device_index)
        self.lib = load_library()
        handle = C.c_void_p()
        _check(self.lib.vstab_create(C.byref(handle), self.device_index), "vstab_create")
        self.handle = handle
        # what the most recent node-boundary transfers did (upload / download below): chunks that crossed PCIe as bytes of
        # all chunks; whether the mask came back as bytes
        self.last_upload_coded = (0, 0)
        self.last_download_coded = False
        self.use_torch_stream()

    def close(self) -> None:
        if getattr(self, "handle", None):
            self.lib.vstab_destroy(self.handle)
            self.handle = None

    def __del__(self):  # pragma: no cover - best effort
        try:
            self.close()
        except Exception:
            pass

    def side_stream(self):
        """A second torch stream of this context's device (created once): for a pass that may run beside the main work."""
        s = getattr(self, "_side_stream", None)
        if s is None:
            s = self._side_stream = self.torch.cuda.Stream(device=self.device)
        return s

    def use_torch_stream(self) -> None:
        stream = self.torch.cuda.current_stream(self.device)
        _check(self.lib.vstab_set_stream(self.handle, C.c_void_p(stream.cuda_stream)), "vstab_set_stream")

    def synchronize(self) -> None:
        _check(self.lib.vstab_synchronize(self.handle), "vstab_synchronize")

    def set_timing(self, enabled: bool, detail: bool = False, warp_only: bool = False) -> None:
        """detail: also bracket the stages inside a DIS call (a measurement pass of its own: see vstab.h).
        warp_only: events around the warp launches only (what a timed loop keeps: an event pair costs the stream ~10 us)."""
        level = 0 if not enabled else (3 if warp_only else (2 if detail else 1))
        _check(self.lib.vstab_set_timing(self.handle, level), "vstab_set_timing")

    def dis_stage_ms(self) -> Dict[str, Any]:
        """Milliseconds of the stages of the last DIS call under set_timing(True, detail=True):
        {"prep": pyramid + coarsest level's preparation, "pis4": {"L5": .., ..}, "level": {..}, "final": ..} -- levels by
        pyramid index (2 = the finest of the default configuration)."""
        def ms(kind):
            out = C.c_float()
            return float(out.value) if self.lib.vstab_last_kernel_ms(self.handle, kind.encode(), C.byref(out)) == 0 else None

        res: Dict[str, Any] = {"prep": ms("dis_prep"), "pis4": {}, "level": {}, "final": ms("dis_final")}
        for lvl in range(16):
            for stage in ("pis4", "level"):
                v = ms(f"dis_{stage}_L{lvl}")
                if v is not None:
                    res[stage][f"L{lvl}"] = round(v, 4)
        for k in ("prep", "final"):
            res[k] = round(res[k], 4) if res[k] is not None else None
        res["pis4_total"] = round(sum(res["pis4"].values()), 4)
        res["level_total"] = round(sum(res["level"].values()), 4)
        return res

    def last_kernel_ms(self, kind: str) -> float:
        out = C.c_float()
        _check(self.lib.vstab_last_kernel_ms(self.handle, kind.encode(), C.byref(out)), "vstab_last_kernel_ms")
        return float(out.value)

    def kernel_ms_stats(self, kind: str):
        """(total ms, launches) of all calls of `kind` since set_timing(True)."""
        total, launches = C.c_double(), C.c_int()
        _check(self.lib.vstab_kernel_ms_stats(self.handle, kind.encode(), C.byref(total), C.byref(launches)), "vstab_kernel_ms_stats")
        return float(total.value), int(launches.value)

    # ------------------------------------------------------------------ helpers
    def _as_device_frames(self, frames):
        torch = self.torch
        if not isinstance(frames, torch.Tensor):
            frames = torch.from_numpy(np.ascontiguousarray(frames, dtype=np.float32))
        if frames.dtype != torch.float32:
            frames = frames.to(torch.float32)
        if frames.device != self.device:
            frames = frames.to(self.device, non_blocking=True)
        return frames.contiguous()

    # ------------------------------------------------------------------ node-boundary transfers
    def upload(self, host_tensor):
        """CPU tensor (pageable) -> new device tensor through the library's pinned ring.  float32 tensors take
        vstab_upload_f32_coded: chunks that hold nothing but float32(k) / 255 -- a ComfyUI IMAGE decoded from 8-bit video --
        cross PCIe as bytes and are expanded on the device to the same bits (VSTAB_XFER_CODED=0: always as float32).
        `last_upload_coded` = (chunks that crossed as bytes, chunks) of the most recent call."""
        torch = self.torch
        src = host_tensor.contiguous()
        dst = torch.empty(src.shape, dtype=src.dtype, device=self.device)
        self.use_torch_stream()
        if src.dtype == torch.float32 and _coded_transfers():
            coded = C.c_size_t(0)
            _check(self.lib.vstab_upload_f32_coded(self.handle, src.data_ptr(), _dev_ptr(dst), src.numel(), C.byref(coded)), "vstab_upload_f32_coded")
            self.last_upload_coded = (int(coded.value), -(-src.numel() // (32 << 20)))
        else:
            _check(self.lib.vstab_upload(self.handle, src.data_ptr(), _dev_ptr(dst), src.numel() * src.element_size()), "vstab_upload")
            self.last_upload_coded = (0, -(-src.numel() // (32 << 20)))
        return dst

    def upload_u8_as_f32(self, host_tensor):
        """uint8 CPU tensor -> new float32 device tensor holding float32(k) / 255 (stabilizer_utils.py:122-126): the bytes cross
        PCIe, the division runs on the device (vstab_upload_u8_as_f32)."""
        torch = self.torch
        src = host_tensor.contiguous()
        if src.dtype != torch.uint8 or src.device.type != "cpu":
            raise VstabError("upload_u8_as_f32: a uint8 CPU tensor is required")
        dst = torch.empty(src.shape, dtype=torch.float32, device=self.device)
        self.use_torch_stream()
        _check(self.lib.vstab_upload_u8_as_f32(self.handle, src.data_ptr(), _dev_ptr(dst), src.numel()), "vstab_upload_u8_as_f32")
        return dst

    def download(self, device_tensor, mask=False, levels=1):
        """Device tensor -> new CPU tensor (pageable, as the reference returns) through the pinned ring (vstab_download).
        mask=True (a float32 padding mask): vstab_download_mask_levels -- a mask of zeros and ones (levels = 1), or the motion-blur
        warp's 1 - c / S (levels = S samples), crosses as bytes and is expanded by the host threads; a mask with any other value
        takes the plain path.  `last_download_coded` tells which."""
        torch = self.torch
        src = device_tensor.contiguous()
        dst = torch.empty(src.shape, dtype=src.dtype)
        self.use_torch_stream()
        if mask and src.dtype == torch.float32 and _coded_transfers():
            coded = C.c_int(0)
            _check(self.lib.vstab_download_mask_levels(self.handle, _dev_ptr(src), dst.data_ptr(), src.numel(), max(1, min(255, int(levels))),
                                                       C.byref(coded)), "vstab_download_mask_levels")
            self.last_download_coded = bool(coded.value)
        else:
            _check(self.lib.vstab_download(self.handle, _dev_ptr(src), dst.data_ptr(), src.numel() * src.element_size()), "vstab_download")
            self.last_download_coded = False
        return dst

    # ------------------------------------------------------------------ warp
    def warp_batch(self, frames, matrices, out_size, interp="bilinear", border=(0.0, 0.0, 0.0),
                   subpix=None, want_mask=True, want_count=False, out=None, out_mask=None):
        """frames [N,H,W,3] f32 (device or host) -> (dst [N,h,w,3], mask [N,h,w] | None, counts [N] | None)."""
        torch = self.torch
        src = self._as_device_frames(frames)
        n, sh, sw, ch = src.shape
        if ch != 3:
            raise VstabError(f"warp_batch expects 3-channel frames, got {ch}")
        out_w, out_h = int(out_size[0]), int(out_size[1])
        m = np.ascontiguousarray(matrices, dtype=np.float32).reshape(n, 9)
        b = np.ascontiguousarray(border, dtype=np.float32).reshape(3)
        dst = out if out is not None else torch.empty((n, out_h, out_w, 3), dtype=torch.float32, device=self.device)
        mask = None
        if want_mask:
            mask = out_mask if out_mask is not None else torch.empty((n, out_h, out_w), dtype=torch.float32, device=self.device)
        counts = torch.empty((n,), dtype=torch.int32, device=self.device) if (want_count and want_mask) else None
        self.use_torch_stream()
        _check(
            self.lib.vstab_warp_batch(
                self.handle, _dev_ptr(src), n, sh, sw, m.ctypes.data, out_h, out_w, INTERP[interp], b.ctypes.data,
                SUBPIX[subpix or DEFAULT_SUBPIX], _dev_ptr(dst), _dev_ptr(mask) if mask is not None else None,
                _dev_ptr(counts) if counts is not None else None,
            ),
            "vstab_warp_batch",
        )
        if counts is not None and out is None:
            counts._vstab_fetch = lambda n=n: self.last_pad_counts(n)   # valid until the next warp with counts of this context
        return dst, mask, counts

    def warp_blur_batch(self, frames, matrices64, out_size, blur, samples, interp="bilinear",
                        border=(0.0, 0.0, 0.0), subpix=None, want_mask=True, out=None, out_mask=None, clip_first=0):
        """frames = frames [clip_first, clip_first + n) of the clip whose motion matrices are `matrices64` [total,3,3]
        (total == n and clip_first == 0 for a whole clip; a shard passes the replicated table of the whole clip)."""
        torch = self.torch
        src = self._as_device_frames(frames)
        n, sh, sw, ch = src.shape
        if ch != 3:
            raise VstabError(f"warp_blur_batch expects 3-channel frames, got {ch}")
        out_w, out_h = int(out_size[0]), int(out_size[1])
        m = np.ascontiguousarray(matrices64, dtype=np.float64).reshape(-1, 9)
        total = m.shape[0]
        if clip_first < 0 or clip_first + n > total:
            raise VstabError(f"warp_blur_batch: frames [{clip_first}, {clip_first + n}) outside a clip of {total} matrices")
        ts = np.ascontiguousarray(np.linspace(0.0, float(blur), int(samples), dtype=np.float64))
        b = np.ascontiguousarray(border, dtype=np.float32).reshape(3)
        dst = out if out is not None else torch.empty((n, out_h, out_w, 3), dtype=torch.float32, device=self.device)
        mask = None
        if want_mask:
            mask = out_mask if out_mask is not None else torch.empty((n, out_h, out_w), dtype=torch.float32, device=self.device)
        self.use_torch_stream()
        _check(
            self.lib.vstab_warp_blur_clip_batch(
                self.handle, _dev_ptr(src), n, sh, sw, m.ctypes.data, total, int(clip_first), ts.ctypes.data, int(samples),
                out_h, out_w, INTERP[interp], b.ctypes.data, SUBPIX[subpix or DEFAULT_SUBPIX], _dev_ptr(dst),
                _dev_ptr(mask) if mask is not None else None,
            ),
            "vstab_warp_blur_clip_batch",
        )
        return dst, mask

    # ------------------------------------------------------------------ estimation
    def gray_downscale(self, frames, work_size, want_range=False):
        """frames [N,H,W,3] f32 -> gray u8 [N,work_h,work_w] (work_size=(w,h) or None for full size).
        want_range: also the per-frame maximum sample (device f32 [N], NaN-propagating) from the same pass."""
        torch = self.torch
        src = self._as_device_frames(frames)
        n, sh, sw, ch = src.shape
        if ch != 3:
            raise VstabError(f"gray_downscale expects 3-channel frames, got {ch}")
        ww, wh = (sw, sh) if work_size is None else (int(work_size[0]), int(work_size[1]))
        gray = torch.empty((n, wh, ww), dtype=torch.uint8, device=self.device)
        self.use_torch_stream()
        if want_range:
            peaks = torch.empty((n,), dtype=torch.float32, device=self.device)
            _check(self.lib.vstab_gray_downscale_range(self.handle, _dev_ptr(src), n, sh, sw, wh, ww, _dev_ptr(gray), _dev_ptr(peaks)),
                   "vstab_gray_downscale_range")
            # the host's copy needs no transfer: the kernel mirrors the maxima into coherent host memory and this fetch waits
            # for that kernel alone (host_math.apply_value_range uses it; valid until the next range pass of this context)
            peaks._vstab_fetch = lambda n=n: self.last_frame_peaks(n)
            return gray, peaks
        _check(self.lib.vstab_gray_downscale(self.handle, _dev_ptr(src), n, sh, sw, wh, ww, _dev_ptr(gray)),
               "vstab_gray_downscale")
        return gray

    def last_pad_counts(self, n: int) -> np.ndarray:
        """Host copy of the padded-pixel counts of the latest warp_batch / warp_batch_planned(want_count=True) over n frames."""
        out = np.empty((int(n),), np.uint32)
        _check(self.lib.vstab_last_pad_counts(self.handle, int(n), out.ctypes.data), "vstab_last_pad_counts")
        return out.astype(np.int64)

    def last_frame_peaks(self, n: int) -> np.ndarray:
        """Host copy of the per-frame maxima of the latest gray_downscale(..., want_range=True) over n frames."""
        out = np.empty((int(n),), np.float32)
        _check(self.lib.vstab_last_frame_peaks(self.handle, int(n), out.ctypes.data), "vstab_last_frame_peaks")
        return out

    def frame_range(self, frames):
        """Per-frame maximum sample of frames [N,H,W,3] f32 on the device (NaN-propagating) -> device f32 [N]."""
        torch = self.torch
        src = self._as_device_frames(frames)
        n, sh, sw, ch = src.shape
        peaks = torch.empty((n,), dtype=torch.float32, device=self.device)
        self.use_torch_stream()
        _check(self.lib.vstab_frame_range(self.handle, _dev_ptr(src), n, sh, sw * ch // 3, _dev_ptr(peaks)), "vstab_frame_range")
        return peaks

    def apply_value_range(self, frames, peaks):
        """stabilizer_utils.py:127-131 on a device clip: a NEW tensor with the frames whose maximum (peaks, device f32 [N])
        exceeds 1.5 divided by 255 in IEEE float32, the others copied."""
        torch = self.torch
        src = self._as_device_frames(frames)
        n, sh, sw, ch = src.shape
        out = torch.empty_like(src)
        peaks = peaks.to(device=self.device, dtype=torch.float32).contiguous()
        self.use_torch_stream()
        _check(self.lib.vstab_apply_value_range(self.handle, _dev_ptr(src), n, sh, sw * ch // 3, _dev_ptr(peaks), _dev_ptr(out)),
               "vstab_apply_value_range")
        return out

    def dis_flow_batch(self, gray, sample_step=8, want_full=False, want_grid=True, clip_start=True):
        """gray u8 [N,h,w] (device) -> (flow [N-1,h,w,2] | None, grid_flow [N-1,gh,gw,2] | None).
        clip_start=False: these frames are a later shard of a clip (see vstab_dis_set_clip_start)."""
        torch = self.torch
        _check(self.lib.vstab_dis_set_clip_start(self.handle, 1 if clip_start else 0), "vstab_dis_set_clip_start")
        if gray.device != self.device:
            gray = gray.to(self.device)
        gray = gray.contiguous()
        n, h, w = gray.shape
        pairs = n - 1
        if pairs < 1:
            raise VstabError("dis_flow_batch needs at least two frames")
        gh, gw = (h + sample_step - 1) // sample_step, (w + sample_step - 1) // sample_step
        flow = torch.empty((pairs, h, w, 2), dtype=torch.float32, device=self.device) if want_full else None
        grid = torch.empty((pairs, gh, gw, 2), dtype=torch.float32, device=self.device) if want_grid else None
        self.use_torch_stream()
        _check(
            self.lib.vstab_dis_flow_batch(
                self.handle, _dev_ptr(gray), n, h, w, _dev_ptr(flow) if flow is not None else None,
                _dev_ptr(grid) if grid is not None else None, int(sample_step),
            ),
            "vstab_dis_flow_batch",
        )
        return flow, grid

    def sample_fit_batch(self, grid_flow, step, requested_mode):
        """grid_flow [P,gh,gw,2] (device) -> structured table [P,3] (FIT_DTYPE), one row per pair and mode."""
        if grid_flow.device != self.device:
            grid_flow = grid_flow.to(self.device)
        grid_flow = grid_flow.contiguous()
        pairs, gh, gw, _ = grid_flow.shape
        table = np.zeros((pairs, 3), FIT_DTYPE)
        self.use_torch_stream()
        _check(
            self.lib.vstab_sample_fit_batch(
                self.handle, _dev_ptr(grid_flow), pairs, gh, gw, int(step), MODES[requested_mode], table.ctypes.data),
            "vstab_sample_fit_batch",
        )
        return table

    def sample_fit_batch_begin(self, grid_flow, step, requested_mode):
        """Launch the fits of sample_fit_batch without waiting: the records stay on the device (fit_records_device) and
        their download is queued; sample_fit_batch_end(pairs) collects the host table.  Work queued in between (the
        device plan, the warp) does not delay that download."""
        if grid_flow.device != self.device:
            grid_flow = grid_flow.to(self.device)
        grid_flow = grid_flow.contiguous()
        pairs, gh, gw, _ = grid_flow.shape
        self.use_torch_stream()
        _check(self.lib.vstab_sample_fit_batch_begin(self.handle, _dev_ptr(grid_flow), pairs, gh, gw, int(step), MODES[requested_mode]),
               "vstab_sample_fit_batch_begin")
        self._fit_grid = grid_flow   # keep the input alive until the kernels have run
        return pairs

    def fit_records_device(self) -> int:
        """Device address of the records of the last sample_fit_batch_begin ([pairs*3] vstab_fit_record)."""
        ptr = self.lib.vstab_fit_records_device(self.handle)
        if not ptr:
            raise VstabError("fit_records_device: no fit is pending")
        return int(ptr)

    def sample_fit_batch_end(self, pairs):
        table = np.zeros((pairs, 3), FIT_DTYPE)
        _check(self.lib.vstab_sample_fit_batch_end(self.handle, int(pairs), table.ctypes.data), "vstab_sample_fit_batch_end")
        self._fit_grid = None
        return table

    # ------------------------------------------------------------------ speculative device plan (F6-F12, crop_and_pad)
    def fit_records_copy(self, dst, pairs):
        """The pending fit's device records into `dst` (device uint8 tensor of >= pairs * 3 records), stream-ordered."""
        self.use_torch_stream()
        _check(self.lib.vstab_fit_records_copy(self.handle, _dev_ptr(dst), int(pairs)), "vstab_fit_records_copy")

    def flow_plan_device(self, records_ptr, pairs, requested_mode, source_size, working_size, smooth, fps, strength, camera_lock,
                         seg_pairs=None, seg_rows=0, warp_frames=0, framing="crop_and_pad"):
        """Queue plan_kernel behind the fits: records (device address) -> the warp's transform table, on the device.
        seg_pairs / seg_rows: the records are an all-gather's receive buffer (seg_rows pairs per rank, seg_pairs[r] valid).
        warp_frames: frames the following warp_batch_planned(want_count=True) will warp -- its count array is allocated here
        and zeroed by the plan kernel, so that no fill launch sits between plan and warp.
        framing: "crop_and_pad" (matrices recentred on the frames' common region) or "expand" (shifted so that the frames'
        union starts at the origin; the canvas size comes from flow_plan_result's region: expand_canvas())."""
        if framing not in ("crop_and_pad", "expand"):
            raise VstabError(f"flow_plan_device: framing {framing!r} is not formed on the device")
        self._planned_counts = None
        up = down = None
        if working_size is not None:
            sx, sy = working_size[0] / float(source_size[0]), working_size[1] / float(source_size[1])
            up = np.array([1.0 / sx, 1.0 / sy, 1.0], np.float64)
            down = np.array([sx, sy, 1.0], np.float64)
        seg = np.ascontiguousarray(seg_pairs, np.int32) if seg_pairs is not None else None
        self.use_torch_stream()
        if warp_frames:
            # registered right in front of the plan call, with every argument already built: the library hands the pointer to
            # THAT call's kernel only (and drops it if the call is refused), so no stale registration can outlive the tensor
            counts = self.torch.empty((int(warp_frames),), dtype=self.torch.int32, device=self.device)
            _check(self.lib.vstab_flow_plan_zero_counts(self.handle, _dev_ptr(counts), int(warp_frames)), "vstab_flow_plan_zero_counts")
            self._planned_counts = counts
        _check(self.lib.vstab_flow_plan_device(
            self.handle, C.c_void_p(int(records_ptr)), int(pairs), MODES[requested_mode],
            up.ctypes.data if up is not None else None, down.ctypes.data if down is not None else None,
            float(smooth), float(fps), float(strength), 1 if camera_lock else 0, int(source_size[0]), int(source_size[1]),
            len(seg) if seg is not None else 0, seg.ctypes.data if seg is not None else None, int(seg_rows),
            1 if framing == "expand" else 0),
            "vstab_flow_plan_device")

    @staticmethod
    def expand_canvas(region):
        """(width, height) of the expand canvas from the device plan's region x_min, y_min, x_max, y_max -- the arithmetic of
        _prepare_expand_transform (stabilizer_utils.py:386-406); None if the region is not finite."""
        import math

        x0, y0, x1, y1 = (float(v) for v in region)
        if not all(math.isfinite(v) for v in (x0, y0, x1, y1)):
            return None
        return max(int(math.ceil(x1 - x0)), 1), max(int(math.ceil(y1 - y0)), 1)

    def flow_plan_result(self, frames, params):
        """(final float32 matrices [frames,3,3], path, target [frames,params], region [4]) of the pending device plan."""
        final = np.zeros((frames, 3, 3), np.float32)
        path = np.zeros((frames, params), np.float64)
        target = np.zeros((frames, params), np.float64)
        region = np.zeros(4, np.float64)
        _check(self.lib.vstab_flow_plan_result(self.handle, int(frames), final.ctypes.data, path.ctypes.data, target.ctypes.data,
                                               region.ctypes.data), "vstab_flow_plan_result")
        return final, path, target, region

    def warp_batch_planned(self, frames, first, out_size, border=(0.0, 0.0, 0.0), subpix=None, want_mask=True, want_count=False):
        """warp_batch (bilinear) for frames [first, first + n) of the clip whose plan is pending on the device."""
        torch = self.torch
        src = self._as_device_frames(frames)
        n, sh, sw, ch = src.shape
        if ch != 3:
            raise VstabError(f"warp_batch_planned expects 3-channel frames, got {ch}")
        out_w, out_h = int(out_size[0]), int(out_size[1])
        b = np.ascontiguousarray(border, dtype=np.float32).reshape(3)
        dst = torch.empty((n, out_h, out_w, 3), dtype=torch.float32, device=self.device)
        mask = torch.empty((n, out_h, out_w), dtype=torch.float32, device=self.device) if want_mask else None
        counts = None
        if want_count and want_mask:
            counts = getattr(self, "_planned_counts", None)      # zeroed by the plan kernel, if the plan call was told of this warp
            self._planned_counts = None
            if counts is None or counts.shape[0] != n:
                counts = torch.empty((n,), dtype=torch.int32, device=self.device)
        self.use_torch_stream()
        _check(self.lib.vstab_warp_batch_planned(
            self.handle, _dev_ptr(src), int(first), n, sh, sw, out_h, out_w, INTERP["bilinear"], b.ctypes.data,
            SUBPIX[subpix or DEFAULT_SUBPIX], _dev_ptr(dst), _dev_ptr(mask) if mask is not None else None,
            _dev_ptr(counts) if counts is not None else None), "vstab_warp_batch_planned")
        if counts is not None:
            counts._vstab_fetch = lambda n=n: self.last_pad_counts(n)   # valid until the next warp with counts of this context
        return dst, mask, counts

    # ------------------------------------------------------------------ Classic estimator (sparse features + LK)
    def gftt_batch(self, gray, max_corners=400, quality=0.01, min_distance=7.0, block_size=21):
        """gray u8 [N,h,w] (device) -> (corners [N,max_corners,2] f32, counts [N] i32), both on the device."""
        torch = self.torch
        if gray.device != self.device:
            gray = gray.to(self.device)
        gray = gray.contiguous()
        n, h, w = gray.shape
        corners = torch.zeros((n, int(max_corners), 2), dtype=torch.float32, device=self.device)
        counts = torch.zeros((n,), dtype=torch.int32, device=self.device)
        self.use_torch_stream()
        _check(self.lib.vstab_gftt_batch(self.handle, _dev_ptr(gray), n, h, w, int(max_corners), float(quality),
                                         float(min_distance), int(block_size), _dev_ptr(corners), _dev_ptr(counts)),
               "vstab_gftt_batch")
        return corners, counts

    def lk_track_batch(self, gray, points, counts, win=31, max_level=3, max_count=50, epsilon=0.01, want_raw=False):
        """Track points[i] (features of frame i) from frame i to frame i+1 for the N-1 pairs of gray [N,h,w].
        Returns point_pairs [N-1,max,4] (next = NaN where untracked) and, with want_raw, (next [N-1,max,2], status)."""
        torch = self.torch
        gray = gray.contiguous()
        n, h, w = gray.shape
        pairs = n - 1
        if pairs < 1:
            raise VstabError("lk_track_batch needs at least two frames")
        points = points[:pairs].contiguous()
        counts = counts[:pairs].contiguous()
        max_pts = int(points.shape[1])
        out = torch.empty((pairs, max_pts, 4), dtype=torch.float32, device=self.device)
        nxt = torch.zeros((pairs, max_pts, 2), dtype=torch.float32, device=self.device) if want_raw else None
        status = torch.zeros((pairs, max_pts), dtype=torch.uint8, device=self.device) if want_raw else None
        self.use_torch_stream()
        _check(self.lib.vstab_lk_track_batch(self.handle, _dev_ptr(gray), n, h, w, _dev_ptr(points), _dev_ptr(counts), max_pts,
                                             int(win), int(max_level), int(max_count), float(epsilon), _dev_ptr(out),
                                             _dev_ptr(nxt) if nxt is not None else None,
                                             _dev_ptr(status) if status is not None else None),
               "vstab_lk_track_batch")
        return (out, nxt, status) if want_raw else out

    def points_fit_batch(self, point_pairs, counts, requested_mode):
        """point_pairs [P,max,4] + counts [P] (device) -> structured table [P,3] (FIT_DTYPE)."""
        point_pairs = point_pairs.contiguous()
        counts = counts[: point_pairs.shape[0]].contiguous()
        pairs, max_pts, _ = point_pairs.shape
        table = np.zeros((pairs, 3), FIT_DTYPE)
        self.use_torch_stream()
        _check(self.lib.vstab_points_fit_batch(self.handle, _dev_ptr(point_pairs), _dev_ptr(counts), pairs, max_pts,
                                               MODES[requested_mode], table.ctypes.data),
               "vstab_points_fit_batch")
        return table

    # ------------------------------------------------------------------ fallback estimator (phase correlation)
    def phase_correlate_batch(self, gray):
        """gray u8 [N,h,w] (device) -> (structured table [N-1,3] (FIT_DTYPE, translation row only),
        shifts f64 [N-1,3] = (tx, ty, response) of cv2.phaseCorrelate for every consecutive pair)."""
        if gray.device != self.device:
            gray = gray.to(self.device)
        gray = gray.contiguous()
        if gray.dtype != self.torch.uint8 or gray.dim() != 3:
            raise ValueError("phase_correlate_batch expects a uint8 [N,h,w] tensor")
        n, h, w = gray.shape
        if n < 2:
            raise ValueError("phase_correlate_batch needs at least two frames")
        table = np.zeros((n - 1, 3), FIT_DTYPE)
        shifts = np.zeros((n - 1, 3), np.float64)
        self.use_torch_stream()
        _check(self.lib.vstab_phase_correlate_batch(self.handle, _dev_ptr(gray), n, h, w, table.ctypes.data, shifts.ctypes.data),
               "vstab_phase_correlate_batch")
        return table, shifts

    def crop_analysis(self, matrices, src_size, out_size):
        """Nearest coverage of every frame -> (bbox [n,4] of the 3x3-closed coverage or -1, AND of all frames eroded 3x3)."""
        m = np.ascontiguousarray(matrices, dtype=np.float32).reshape(-1, 9)
        n = m.shape[0]
        sw, sh = int(src_size[0]), int(src_size[1])
        ow, oh = int(out_size[0]), int(out_size[1])
        bbox = np.zeros((n, 4), np.int32)
        common = np.zeros((oh, ow), np.uint8)
        self.use_torch_stream()
        _check(self.lib.vstab_crop_analysis(self.handle, m.ctypes.data, n, sh, sw, oh, ow, bbox.ctypes.data, common.ctypes.data),
               "vstab_crop_analysis")
        return bbox, common

    def common_coverage(self, matrices, src_size, out_size):
        """AND over frames of the nearest coverage (motion_apply.py:205-227) -> bool [out_h,out_w] on the host."""
        m = np.ascontiguousarray(matrices, dtype=np.float32).reshape(-1, 9)
        sw, sh = int(src_size[0]), int(src_size[1])
        ow, oh = int(out_size[0]), int(out_size[1])
        common = np.zeros((oh, ow), np.uint8)
        self.use_torch_stream()
        _check(self.lib.vstab_common_coverage(self.handle, m.ctypes.data, m.shape[0], sh, sw, oh, ow, common.ctypes.data),
               "vstab_common_coverage")
        return common.astype(bool)

    def trajectory(self, deltas, smooth, fps, strength, camera_lock):
        d = np.ascontiguousarray(deltas, dtype=np.float64)
        n = d.shape[0] + 1
        p = d.shape[1]
        path = np.zeros((n, p), np.float64)
        target = np.zeros((n, p), np.float64)
        _check(
            self.lib.vstab_trajectory(self.handle, d.ctypes.data, n, p, float(smooth), float(fps), float(strength),
                                      1 if camera_lock else 0, path.ctypes.data, target.ctypes.data),
            "vstab_trajectory",
        )
        return path, target


_default_ctx: dict = {}


def default_context() -> Context:
    """Per-process context on the current torch device."""
    import torch

    if not torch.cuda.is_available():
        raise VstabError("no MI355X / HIP device visible: the stabilizer hot path has no CPU fallback")
    idx = torch.cuda.current_device()
    ctx = _default_ctx.get(idx)
    if ctx is None:
        ctx = Context(idx)
        _default_ctx[idx] = ctx
    return ctx
