"""cProfile of one Flow pass on the fallback estimator (where does the host time go)."""
import cProfile
import os
import pstats
import sys
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
ctx = native.Context(0)
frames = bench.synth_clip(n, 0, h, w, torch.device("cuda", 0))
torch.cuda.synchronize()


def step():
    context = hm.VideoContext([None] * n, hm.FrameAdapter(np.dtype(np.float32), False, "0_1", "torch", False), w, h, 3, None, "sequence", {},
                              batch=frames)
    return fp._stabilize_frames(context, "crop_and_pad", "similarity", False, 0.7, 0.5, 0.6, (127, 127, 127), 16.0, ctx=ctx, keep_on_device=True)


for _ in range(2):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    step()
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
