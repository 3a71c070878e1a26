"""Workload for a kernel trace of DIS on a short clip (a multi-GPU rank's share): 129 frames -> 128 pairs, 5 calls."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import __graft_entry__ as graft
graft.load_package()
import bench
from vstab_amd import native
ctx = native.Context(0)
frames = bench.synth_clip(129, 0, 1080, 1920, torch.device("cuda", 0))
gray = ctx.gray_downscale(frames, (960, 540))
for _ in range(5):
    _, g = ctx.dis_flow_batch(gray, sample_step=8)
    ctx.sample_fit_batch(g, 8, "similarity")
torch.cuda.synchronize()
