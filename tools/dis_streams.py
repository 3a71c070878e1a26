"""Does estimating the two halves of a short clip on two HIP streams (two contexts, own scratch) beat one call?
At few pairs the coarse pyramid levels are latency-bound and leave the chip idle; a second stream could fill it.
Prints wall ms (host timer around launch + sync, median of 8) for one call of N pairs vs two concurrent calls of N/2."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
import __graft_entry__ as graft
graft.load_package()
import bench
from vstab_amd import native

dev = torch.device("cuda", 0)
a, b = native.Context(0), native.Context(0)
frames = bench.synth_clip(257, 0, 1080, 1920, dev)
gray = a.gray_downscale(frames, (960, 540))
del frames
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def one(n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    _, g = a.dis_flow_batch(gray[:n], sample_step=8)
    a.sample_fit_batch(g, 8, "similarity")
    return (time.perf_counter() - t0) * 1e3


def two(n):
    h = n // 2
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with torch.cuda.stream(s1):
        _, g1 = a.dis_flow_batch(gray[: h + 1], sample_step=8)
    with torch.cuda.stream(s2):
        _, g2 = b.dis_flow_batch(gray[h:n], sample_step=8, clip_start=False)
    with torch.cuda.stream(s1):
        a.sample_fit_batch(g1, 8, "similarity")
    with torch.cuda.stream(s2):
        b.sample_fit_batch(g2, 8, "similarity")
    return (time.perf_counter() - t0) * 1e3


for n in (65, 129, 257):
    for fn in (one, two):
        for _ in range(3): fn(n)
        ts = [fn(n) for _ in range(8)]
        print(f"{n - 1:4d} pairs  {fn.__name__}: {np.median(ts):.3f} ms (min {min(ts):.3f})")
