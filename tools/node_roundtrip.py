"""PCIe-inclusive rate of the Flow node as ComfyUI calls it: CPU tensor in, CPU tensors out."""
import sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import __graft_entry__ as graft
graft.load_package()
import bench
from vstab_amd import nodes
n, h, w = 256, 1080, 1920
frames = bench.synth_clip(n, 0, h, w, torch.device("cuda", 0)).cpu()
for r in range(3):
    t0 = time.perf_counter()
    out = nodes.VideoStabilizerFlow.execute(frames, 16.0, "crop_and_pad", "similarity", False, 0.7, 0.5, 0.6, "#7F7F7F")
    dt = time.perf_counter() - t0
    print(f"node call (host in / host out): {dt*1e3:.0f} ms -> {n/dt:.0f} frames/s; out {tuple(out[0].shape)} on {out[0].device}")
    del out
