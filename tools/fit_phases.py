"""Fit-kernel time with the kernel cut short after each phase (diagnostic builds made with -DFIT_STOP=1|2|3 from a copy of
vstab_fit.hip that returns early; see the kernel's header comment for the numbers).  The libraries are not kept in the tree."""
import os, subprocess, sys
CHILD = r'''
import sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import __graft_entry__ as graft
graft.load_package()
from vstab_amd import native
import bench
ctx = native.Context(0); ctx.set_timing(True)
frames = bench.synth_clip(256, 0, 1080, 1920, torch.device("cuda", 0))
gray = ctx.gray_downscale(frames, (960, 540))
_, grid = ctx.dis_flow_batch(gray, sample_step=8, want_full=False, want_grid=True)
ts = []
for r in range(8):
    try:
        ctx.sample_fit_batch(grid, 8, "similarity")
    except Exception as e:
        pass
    torch.cuda.synchronize()
    if r >= 3: ts.append(ctx.last_kernel_ms("fit"))
print(round(float(np.median(ts)), 4))
'''
for lib in ["", "fitstop1.so", "fitstop2.so", "fitstop3.so", "fit256.so"]:
    env = dict(os.environ)
    if lib: env["VSTAB_LIB"] = "/root/repo/comfyui-video-stabilizer_amd/lib/ab/" + lib
    out = subprocess.run([sys.executable, "-c", CHILD], capture_output=True, text=True, env=env)
    print(lib or "full(1024)", out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-500:])
