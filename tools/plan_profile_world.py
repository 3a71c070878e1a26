"""cProfile of the replicated host work of an 8-rank run (plan + meta over 2048 frames) on the GPU box."""
import cProfile, pstats, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
import __graft_entry__ as graft
graft.load_package()
import bench
from vstab_amd import flow_pipeline as fp, host_math as hm, native
ctx = native.Context(0)
n, h, w = 256, 1080, 1920
frames = bench.synth_clip(n, 0, h, w, torch.device("cuda", 0))
table = fp.estimate_transitions(ctx, frames, hm._working_estimation_size(w, h), "similarity")
total = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
big = np.concatenate([table] * 9)[: total - 1]
args = (ctx, big, (w, h), total, "crop_and_pad", "similarity", False, 0.7, 0.5, 0.6, (127, 127, 127), 16.0, 16.0)
for _ in range(10):
    plan = fp.plan_stabilization(*args); fp.prepare_meta(plan)
for name, fn in (("plan", lambda: fp.plan_stabilization(*args)), ("meta", lambda: fp.prepare_meta(plan))):
    t0 = time.perf_counter()
    for _ in range(50): fn()
    print(f"{name}: {(time.perf_counter() - t0) / 50 * 1e3:.3f} ms per call at {total} frames")
    pr = cProfile.Profile(); pr.enable()
    for _ in range(50): fn()
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(14)
