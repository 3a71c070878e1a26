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
ap.add_argument("--device-plan", type=int, default=1, help="1: the round-4 flow (plan formed on the device behind the gathered table, "
                "host plan + verification while the warp runs); 0: the host-plan flow of rounds 1-3")
ap.add_argument("--gc", choices=("default", "freeze", "off"), default="default",
                help="host-runtime experiment: gc.freeze() after the warm-up steps / the cyclic collector off during the timed steps")
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

REC = 3 * native.FIT_DTYPE.itemsize

def step_device_plan(tm):
    """Rank 0 of distributed._stabilize_sharded_device_plan with the all-gather replaced by W device-side copies of the
    local records into the receive buffer (rank r > 0 has one more pair: its halo pair, emulated by repeating pair 0)."""
    t0 = time.perf_counter()
    gray = ctx.gray_downscale(frames, work)
    _, grid = ctx.dis_flow_batch(gray, sample_step=fp.SAMPLE_STEP, want_full=False, want_grid=True)
    pairs = ctx.sample_fit_batch_begin(grid, fp.SAMPLE_STEP, "similarity")
    per_rank = [pairs] + [pairs + 1] * (W - 1)
    rows = pairs + 1
    flat = torch.empty((W, rows, REC), dtype=torch.uint8, device=dev)
    ctx.fit_records_copy(flat[0], pairs)
    for r in range(1, W):                       # stands in for the collective: same bytes landing in the same places
        flat[r, 1:pairs + 1].copy_(flat[0, :pairs])
        flat[r, 0].copy_(flat[0, 0])
    host_t = torch.empty(flat.shape, dtype=torch.uint8, pin_memory=True)
    host_t.copy_(flat, non_blocking=True)
    gathered = torch.cuda.Event(); gathered.record()
    ctx.flow_plan_device(flat.data_ptr(), total - 1, "similarity", size, work, 0.5, 16.0, 0.7, False, seg_pairs=per_rank, seg_rows=rows, warp_frames=n)
    dst, mask, counts = ctx.warp_batch_planned(frames, 0, size, border=hm.border_value((127, 127, 127)), want_mask=True, want_count=True)
    t1 = time.perf_counter()
    ctx.sample_fit_batch_end(pairs)
    gathered.synchronize()
    host = host_t.numpy()
    records = np.ascontiguousarray(np.concatenate([host[r, :per_rank[r]] for r in range(W)], axis=0)).view(native.FIT_DTYPE).reshape(-1, 3)
    t2 = time.perf_counter()
    plan = fp.plan_stabilization(ctx, records, size, total, "crop_and_pad", "similarity", False, 0.7, 0.5, 0.6, (127, 127, 127), 16.0, 16.0)
    final_dev = ctx.flow_plan_result(total, 4)[0]
    sub = fp.FlowPlan(plan.final_matrices[:n], plan.output_size, {}, {}, {}, plan.framing_mode, size, 16.0)
    bad = fp._rewarp_mismatched(ctx, frames, sub, final_dev[:n], dst, mask, counts, (127, 127, 127))
    t3 = time.perf_counter()
    meta = fp.prepare_meta(plan)
    t4 = time.perf_counter()
    c = fp._counts_to_host(counts, mirrored=(bad == 0))
    t5 = time.perf_counter()
    meta = fp.complete_meta(meta, plan, np.tile(c, W))
    t6 = time.perf_counter()
    tm.append([t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5])
    assert bad == 0, bad
    return dst, mask, meta

if args.device_plan:
    step = step_device_plan
tm = []
for _ in range(3):
    step(tm)
torch.cuda.synchronize()
import gc
if args.gc == "freeze":
    gc.collect(); gc.freeze()
elif args.gc == "off":
    gc.collect(); gc.disable()
tm = []
t0 = time.perf_counter()
for _ in range(args.steps):
    out = step(tm); del out
torch.cuda.synchronize()
el = (time.perf_counter() - t0) / args.steps
a = np.mean(np.array(tm), 0) * 1e3
if args.device_plan:
    print(f"world {W} (device plan): {el*1e3:.2f} ms/step -> {n*W/el:.0f} frames/s aggregate if all ranks match; host: launch everything "
          f"{a[0]:.2f}, wait fits + gathered table D2H {a[1]:.2f}, host plan + verify {a[2]:.2f}, prepare_meta {a[3]:.2f}, wait warp {a[4]:.2f}, "
          f"complete {a[5]:.2f} ms")
else:
    print(f"world {W}: {el*1e3:.2f} ms/step -> {n*W/el:.0f} frames/s aggregate if all ranks match; "
          f"estimate(sync) {a[0]:.2f}, plan {a[1]:.2f}, warp launch {a[2]:.2f}, prepare_meta {a[3]:.2f}, wait warp {a[4]:.2f}, complete {a[5]:.2f} ms")
