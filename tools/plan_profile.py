"""cProfile of plan_stabilization alone (256 transitions) on the GPU box, 200 repetitions."""
import cProfile, pstats, sys
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
args = (ctx, table, (w, h), n, "crop_and_pad", "similarity", False, 0.7, 0.5, 0.6, (127, 127, 127), 16.0, 16.0)
for _ in range(20): fp.plan_stabilization(*args)
pr = cProfile.Profile(); pr.enable()
for _ in range(200): fp.plan_stabilization(*args)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(16)
