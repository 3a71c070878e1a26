"""Host timeline between the end of the estimation and the launch of the warp (the stretch of a Flow step in which the
GPU idles), by re-running the pieces of flow_pipeline._stabilize_frames with timers:  python tools/gap_timeline.py"""
import sys, time
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
work = hm._working_estimation_size(w, h)
acc = {}
def lap(name, t0):
    t1 = time.perf_counter()
    acc[name] = acc.get(name, 0.0) + (t1 - t0)
    return t1
REPS = 60
for rep in range(REPS + 5):
    if rep == 5: acc.clear()
    context = hm._normalize_video_input(frames)
    peaks = []
    t = time.perf_counter()
    table = fp.estimate_transitions(ctx, context.device_batch(ctx), work, "similarity", peaks_out=peaks)
    t = lap("0 estimate (gray+DIS+fit, sync)", t)
    hm.resolve_value_range(context, peaks[0], ctx)
    t = lap("1 value range (peaks D2H)", t)
    mats, modes, confs, resids, active = fp.select_transitions(table, "similarity")
    t = lap("2 select_transitions", t)
    full, dp = native.transitions_to_params(mats, "similarity", (w, h), work)
    t = lap("3+4 rescale + matrices_to_params (one library call)", t)
    path, target = ctx.trajectory(dp, 0.5, 16.0, 0.7, False)
    t = lap("5 trajectory (H2D, kernel, D2H, sync)", t)
    am = native.params_to_matrices(target - path, "similarity")
    t = lap("6 params_to_matrices (library call)", t)
    mins, maxs = native.bounding_boxes(am, w, h)
    ratio = hm._min_content_ratio(mins, maxs, w, h)
    t = lap("7 bounding boxes + ratio", t)
    x0, y0 = float(np.max(mins[:, 0])), float(np.max(mins[:, 1])); x1, y1 = float(np.min(maxs[:, 0])), float(np.min(maxs[:, 1]))
    shift = np.array([[1.0, 0.0, w * 0.5 - (x0 + x1) * 0.5], [0.0, 1.0, h * 0.5 - (y0 + y1) * 0.5], [0.0, 0.0, 1.0]], dtype=np.float32)
    final = np.matmul(shift, am)
    t = lap("8 recentre", t)
    dst, mask, counts = ctx.warp_batch(context.device_batch(ctx), final, (w, h), interp="bilinear", border=hm.border_value((127, 127, 127)), want_mask=True, want_count=True)
    t = lap("9 warp_batch call (alloc, invert, H2D, launch)", t)
    counts.cpu()
    t = lap("10 wait for the warp", t)
for k in sorted(acc, key=lambda s: int(s.split()[0].split("+")[0])):
    print(f"{k:48s} {acc[k] / REPS * 1e6:8.1f} us")
print(f"{'total':48s} {sum(acc.values()) / REPS * 1e6:8.1f} us")
