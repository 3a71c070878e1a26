"""vstab_trajectory call time (host view: stage, kernel, D2H, sync) for clip lengths of 1-GPU and 8-GPU runs."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
import __graft_entry__ as graft
graft.load_package()
from vstab_amd import native
ctx = native.Context(0)
for n in (256, 1024, 2048):
    d = np.random.default_rng(0).normal(0, 1, (n - 1, 4))
    for _ in range(5): ctx.trajectory(d, 0.5, 16.0, 0.7, False)
    t0 = time.perf_counter()
    for _ in range(100): ctx.trajectory(d, 0.5, 16.0, 0.7, False)
    print(f"n={n}: {(time.perf_counter() - t0) * 1e4:.1f} us per call")
