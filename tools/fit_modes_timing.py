"""Fit-stage kernel time (HIP events, `fit` timer of the library) per requested mode on the bench clip's sampled flow:
     python tools/fit_modes_timing.py [lib.so ...]"""
import os, subprocess, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
CHILD = r'''
import sys
sys.path.insert(0, %r)
import numpy as np, torch
import __graft_entry__ as graft
graft.load_package()
from vstab_amd import native
import bench
ctx = native.Context(0); ctx.set_timing(True)
frames = bench.synth_clip(256, 0, 1080, 1920, torch.device("cuda", 0))
gray = ctx.gray_downscale(frames, (960, 540))
_, grid = ctx.dis_flow_batch(gray, sample_step=8, want_full=False, want_grid=True)
out = {}
for mode in ("translation", "similarity", "perspective"):
    ts = []
    for r in range(8):
        t = ctx.sample_fit_batch(grid, 8, mode); torch.cuda.synchronize()
        if r >= 3: ts.append(ctx.last_kernel_ms("fit"))
    out[mode] = (round(float(np.median(ts)), 4), float(np.abs(t["matrix"]).sum()))
print(out)
'''
for lib in (sys.argv[1:] or [""]):
    env = dict(os.environ)
    if lib: env["VSTAB_LIB"] = str(ROOT / lib)
    o = subprocess.run([sys.executable, "-c", CHILD % str(ROOT)], capture_output=True, text=True, env=env)
    print(lib or "default", o.stdout.strip().splitlines()[-1] if o.stdout.strip() else o.stderr[-600:])
