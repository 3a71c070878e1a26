"""Gray + area (+ value-range maxima) kernel time for several workgroup sizes, on the bench clip (256 x 1080p, 2x2 boxes)
and a 4K batch (4x4 boxes):  python tools/gray_forms.py   (the load forms it once compared: profiles/r02_gray_forms.md)"""
import os, subprocess, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
CHILD = r'''
import os, sys
sys.path.insert(0, %r)
import numpy as np, torch
import __graft_entry__ as graft
graft.load_package()
from vstab_amd import native
import bench
ctx = native.Context(0); ctx.set_timing(True)
out = {}
for tag, (n, h, w) in {"1080p": (256, 1080, 1920), "4k": (48, 2160, 3840)}.items():
    frames = bench.synth_clip(n, 0, h, w, torch.device("cuda", 0))
    ts = []
    for r in range(8):
        g, peaks = ctx.gray_downscale(frames, (960, 540), want_range=True); torch.cuda.synchronize()
        if r >= 3: ts.append(ctx.last_kernel_ms("gray"))
    out[tag] = (round(float(np.median(ts)), 4), int(g.to(torch.int64).sum().item()), float(peaks.max().item()))
    del frames
print(out)
'''
for threads in ("", "256", "384", "512", "640", "960"):   # "" = the library's own choice
    env = dict(os.environ)
    if threads: env["VSTAB_GRAY_THREADS"] = threads
    out = subprocess.run([sys.executable, "-c", CHILD % str(ROOT)], capture_output=True, text=True, env=env)
    print(f"threads {threads or 'auto'}:", out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-600:], flush=True)
