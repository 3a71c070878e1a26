"""DIS stage time (HIP events around vstab_dis_flow_batch) for the fused and the split launch form at several clip
lengths:  python tools/dis_modes.py [lib.so ...]   (default: the built library)"""
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
if len(sys.argv) > 1: native.LIB_PATH = __import__("pathlib").Path(sys.argv[1]).resolve()
import bench
ctx = native.Context(0); ctx.set_timing(True)
frames = bench.synth_clip(256, 0, 1080, 1920, torch.device("cuda", 0))
gray = ctx.gray_downscale(frames, (960, 540))
out = {}
for n in (65, 129, 193, 256):
    for split in ("0", "1", "2"):
        os.environ["VSTAB_DIS_SPLIT"] = split
        ts = []
        for r in range(6):
            ctx.dis_flow_batch(gray[:n], sample_step=8); torch.cuda.synchronize()
            if r >= 2: ts.append(ctx.last_kernel_ms("dis"))
        out[(n, split)] = round(float(np.median(ts)), 3)
print(sys.argv[1] if len(sys.argv) > 1 else "default", {f"{n}f/{ {'0': 'fused', '1': 'split-finest', '2': 'split-all'}[s]}": v for (n, s), v in out.items()})
'''
libs = sys.argv[1:] or [""]
for lib in libs:
    cmd = [sys.executable, "-c", CHILD % str(ROOT)] + ([str(ROOT / lib)] if lib else [])
    out = subprocess.run(cmd, capture_output=True, text=True)
    print(out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-600:])
