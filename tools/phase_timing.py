"""Times the Flow pipeline on its fallback estimator (VSTAB_FLOW_BACKEND=phase_correlate) on the C2 clip."""
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
os.environ["VSTAB_FLOW_BACKEND"] = "phase_correlate"

import numpy as np
import torch

import __graft_entry__ as graft
import bench

graft.load_package()
from vstab_amd import flow_pipeline as fp
from vstab_amd import host_math as hm
from vstab_amd import native

n, h, w = 256, 1080, 1920
dev = torch.device("cuda", 0)
ctx = native.Context(0)
ctx.set_timing(True)
frames = bench.synth_clip(n, 0, h, w, dev)
torch.cuda.synchronize()


def step():
    context = hm.VideoContext([None] * n, hm.FrameAdapter(np.dtype(np.float32), False, "0_1", "torch", False), w, h, 3, None, "sequence", {},
                              batch=frames)
    return fp._stabilize_frames(context, "crop_and_pad", "similarity", False, 0.7, 0.5, 0.6, (127, 127, 127), 16.0, ctx=ctx, keep_on_device=True)


for _ in range(2):
    step()
torch.cuda.synchronize()
ctx.set_timing(True)
t0 = time.perf_counter()
for _ in range(5):
    res = step()
    meta = res.meta
    del res          # as bench.py: outputs of the previous pass are released before the next one allocates
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 5
stages = {k: ctx.kernel_ms_stats(k)[0] / max(ctx.kernel_ms_stats(k)[1], 1) for k in ("gray", "phase", "warp")}
print(json.dumps({"backend": meta["flow_backend"], "ms_per_clip": round(dt * 1e3, 3), "frames_per_s": round(n / dt, 1),
                  "stage_ms": {k: round(v, 3) for k, v in stages.items()},
                  "modes": sorted(set(t["mode"] for t in meta["estimated_motion"]["per_transition"]))}))
