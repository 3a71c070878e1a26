"""Cost of the replicated host work of the sharded pipeline at world size W, measured on ONE GPU.

Rank 0's step of `distributed.stabilize_sharded` is replayed with the gathered tables of a W-rank run
synthesised by tiling the local records (no collective is issued; RCCL's share is measured separately by
`bench.py --force-dist`).  Prints per-step wall time and the plan / meta split, i.e. what weak scaling
loses to the O(W) host logic that every rank repeats.
"""
import argparse, sys, time
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import __graft_entry__ as graft
graft.load_package()
import bench
from vstab_amd import flow_pipeline as fp, host_math as hm, native

ap = argparse.ArgumentParser(); ap.add_argument("--world", type=int, default=8); ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--frames", type=int, default=256)
args = ap.parse_args()
dev = torch.device("cuda", 0)
ctx = native.Context(0)
n, h, w, W = args.frames, 1080, 1920, args.world
frames = bench.synth_clip(n, 0, h, w, dev)
total = n * W
size = (w, h)
work = hm._working_estimation_size(w, h)

def step(tm):
    t0 = time.perf_counter()
    local = fp.estimate_transitions(ctx, frames, work, "similarity")
    t1 = time.perf_counter()
    reps = [local] + [np.concatenate([local[:1], local])] * (W - 1)      # other ranks: 256 transitions each (halo pair)
    records = np.concatenate(reps)[: total - 1]
    plan = fp.plan_stabilization(ctx, records, size, total, "crop_and_pad", "similarity", False, 0.7, 0.5, 0.6, (127, 127, 127), 16.0, 16.0)
    t2 = time.perf_counter()
    mats = np.ascontiguousarray(plan.final_matrices[:n], dtype=np.float32)
    dst, mask, counts = ctx.warp_batch(frames, mats, plan.output_size, interp="bilinear", border=hm.border_value((127, 127, 127)),
                                       want_mask=True, want_count=True)
    t3 = time.perf_counter()
    meta = fp.prepare_meta(plan)
    t4 = time.perf_counter()
    c = counts.cpu().numpy()
    t5 = time.perf_counter()
    meta = fp.complete_meta(meta, plan, np.tile(c, W))
    t6 = time.perf_counter()
    tm.append([t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5])
    return dst, mask, meta

tm = []
for _ in range(3):
    step(tm)
torch.cuda.synchronize()
tm = []
t0 = time.perf_counter()
for _ in range(args.steps):
    out = step(tm); del out
torch.cuda.synchronize()
el = (time.perf_counter() - t0) / args.steps
a = np.mean(np.array(tm), 0) * 1e3
print(f"world {W}: {el*1e3:.2f} ms/step -> {n*W/el:.0f} frames/s aggregate if all ranks match; "
      f"estimate(sync) {a[0]:.2f}, plan {a[1]:.2f}, warp launch {a[2]:.2f}, prepare_meta {a[3]:.2f}, wait warp {a[4]:.2f}, complete {a[5]:.2f} ms")
