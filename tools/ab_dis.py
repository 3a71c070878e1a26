"""A/B timing of two builds of libvstab.so inside ONE gpurun call (box-to-box variance is ~5 %).

    python tools/ab_dis.py libA.so libB.so      # paths relative to the repo root

Each library is loaded in its own child process; prints the DIS / fit / gray / warp stage times of the C2 clip.
"""
import subprocess, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
CHILD = r'''
import sys, shutil
sys.path.insert(0, %r)
import numpy as np, torch
import __graft_entry__ as graft
graft.load_package()
from vstab_amd import native
native.LIB_PATH = __import__("pathlib").Path(sys.argv[1]).resolve()
import bench
from vstab_amd import flow_pipeline as fp, host_math as hm
ctx = native.Context(0); ctx.set_timing(True)
frames = bench.synth_clip(256, 0, 1080, 1920, torch.device("cuda", 0))
t = {k: [] for k in ("gray", "dis", "fit", "warp")}
for r in range(8):
    c = hm.VideoContext([None] * 256, hm.FrameAdapter(np.dtype(np.float32), False, "0_1", "torch", False), 1920, 1080, 3, None, "sequence", {}, batch=frames)
    res = fp._stabilize_frames(c, "crop_and_pad", "similarity", False, 0.7, 0.5, 0.6, (127, 127, 127), 16.0, ctx=ctx, keep_on_device=True)
    if r >= 2:
        for k in t: t[k].append(ctx.last_kernel_ms(k))
    del res
print(sys.argv[1], {k: round(float(np.median(v)), 3) for k, v in t.items()})
'''
for _ in range(2):
    for lib in sys.argv[1:]:
        out = subprocess.run([sys.executable, "-c", CHILD % str(ROOT), str(ROOT / lib)], capture_output=True, text=True)
        print(out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-400:])
