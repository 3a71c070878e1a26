"""cProfile of one crop-mode Flow pass (256 x 1080p) on the GPU box: where the keep_fov solver spends its time."""
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
mk = lambda: hm.VideoContext([None] * n, hm.FrameAdapter(np.dtype(np.float32), False, "0_1", "torch", False), w, h, 3, None, "sequence", {}, batch=frames)
fp._stabilize_frames(mk(), "crop", "similarity", False, 0.7, 0.5, 0.6, (127, 127, 127), 16.0, ctx=ctx, keep_on_device=True)
pr = cProfile.Profile(); pr.enable()
fp._stabilize_frames(mk(), "crop", "similarity", False, 0.7, 0.5, 0.6, (127, 127, 127), 16.0, ctx=ctx, keep_on_device=True)
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
