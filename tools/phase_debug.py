"""Developer tool: stage-by-stage GPU vs oracle comparison of the phase-correlation estimator.
Needs lib/libvstab_phasedbg.so = a build with EXTRA=-DVSTAB_PHASE_DEBUG (VSTAB_LIB points at it)."""
import ctypes as C
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import numpy as np
import torch

import __graft_entry__ as graft

graft.load_package()
from oracle import oracle as o
from vstab_amd import native

native.LIB_PATH = ROOT / "comfyui-video-stabilizer_amd" / "lib" / "libvstab_phasedbg.so"
from tests.test_phase_gpu import textured_clip

o.build()
ctx = native.default_context()
n, h, w = 4, 135, 240
gray = textured_clip(n, h, w, seed=h * 7 + w)
_, shifts = ctx.phase_correlate_batch(torch.from_numpy(gray))
M, N = o.optimal_dft_size(h), o.optimal_dft_size(w)
nh = N // 2 + 1
spec = np.zeros((M, nh, 2), np.float32)
surf = np.zeros((M, N), np.float32)
fn = ctx.lib.vstab_phase_debug_dump
fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
colinv = np.zeros((M, nh, 2), np.float32)
assert fn(ctx.handle, n, h, w, spec.ctypes.data, surf.ctypes.data, colinv.ctypes.data) == 0
ref_spec = np.zeros_like(spec)
f2 = o.lib().vo_phase_spectrum
f2.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
f2.restype = None
f2(gray[0].ctypes.data, h, w, ref_spec.ctypes.data)
ref_shifts, ref_surf = o.phase_correlate_clip(gray, want_surface=True)
d = spec != ref_spec
print("spectrum mismatches", int(d.sum()), "of", d.size, "first", np.argwhere(d)[:8].tolist())
if d.any():
    i = tuple(np.argwhere(d)[0])
    print("  gpu", spec[i], "ref", ref_spec[i])
    rows = np.unique(np.argwhere(d)[:, 0]); cols = np.unique(np.argwhere(d)[:, 1])
    print("  rows", rows[:20].tolist(), "cols", cols[:20].tolist())
d2 = surf != ref_surf
print("surface mismatches", int(d2.sum()), "of", d2.size, "max abs", float(np.abs(surf - ref_surf).max()))
print("shifts equal", np.array_equal(shifts, ref_shifts))

f3 = o.lib().vo_phase_debug_pair
f3.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
f3.restype = None
ref_G = np.zeros((M, nh, 2), np.float32)
ref_R = np.zeros((M, N), np.float32)
sh = np.zeros(3)
pair = np.ascontiguousarray(gray[:2])
f3(pair.ctypes.data, h, w, ref_G.ctypes.data, ref_R.ctypes.data, sh.ctypes.data)
d3 = colinv != ref_G
print("column-inverse mismatches", int(d3.sum()), "of", d3.size, "cols", np.unique(np.argwhere(d3)[:, 1])[:30].tolist())
if d3.any():
    i = tuple(np.argwhere(d3)[0]); print("  first", i, colinv[i], ref_G[i])
print("ref surfaces equal", np.array_equal(ref_R, ref_surf))
