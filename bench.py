#!/usr/bin/env python3
"""Headline benchmark: stabilized frames/s of the Video Stabilizer Flow hot path on MI355X.

Workload (BASELINE.json configs[1], "C2"): 256 synthetic 1080p frames per GPU, Flow node defaults
(DIS flow -> similarity fit -> box-smoothed trajectory, strength 0.7 / smooth 0.5 / 16 fps ->
crop_and_pad warp with padding mask).  One "step" = one full pass of the hot path over the clip
with the input frames already resident in HBM and the outputs left in HBM:
    gray+downscale -> DIS (255 pairs) -> fit -> [all-gather of fit records if N > 1] ->
    trajectory -> framing -> warp + mask + padding counts.
N > 1: one process per GPU (torchrun), the clip is 256*N frames sharded contiguously with a
1-frame halo; weak scaling (per-GPU work fixed).  value = frames of all ranks / max-over-ranks time.

Also on the JSON line:
  roofline     - the warp kernel (dominant HBM stream): algorithmic 28 B per output pixel x pixels
                 per launch / average launch time from HIP events on the launch stream
  cpu_baseline - the CPU oracle (a C restatement of the reference's OpenCV path; "port") timed on
                 this host's cores over a bounded sample of the same clip (rank 0, N=1 only)
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec
WARP_BYTES_PER_PIXEL = 28  # read 12 B source + write 12 B frame + 4 B mask


def camera_matrices(n: int, offset: int, width: int, height: int) -> np.ndarray:
    """Per-frame camera transform: the reference's 'handheld' shake recipe (seed 0, 16 fps; committed
    as data in tests/golden/) tiled over the clip, composed with a slow 0.7 px/frame pan."""
    blk = json.loads((ROOT / "tests" / "golden" / "shake_c3_256x1080p.json").read_text())
    base = np.array([e["matrix"] for e in blk["per_frame"]], np.float64)
    sx, sy = width / 1920.0, height / 1080.0
    out = np.empty((n, 3, 3), np.float64)
    for k in range(n):
        i = offset + k
        m = base[i % len(base)].copy()
        m[0, 2] *= sx
        m[1, 2] *= sy
        pan = np.array([[1, 0, 0.7 * i * sx], [0, 1, 0.15 * i * sy], [0, 0, 1.0]])
        out[k] = pan @ m
    return out


def synth_clip(n: int, offset: int, height: int, width: int, device, seed: int = 1234):
    """Band-limited procedural texture sampled analytically under the camera path (no interpolation):
    frame_i(p) = T(M_i^-1 p).  Returns float32 [n,H,W,3] in [0.05,0.95] on `device`."""
    import torch

    rng = np.random.default_rng(seed)
    k = 20
    fx = torch.tensor(rng.uniform(-0.11, 0.11, k) * (1920.0 / width), device=device, dtype=torch.float32)
    fy = torch.tensor(rng.uniform(-0.11, 0.11, k) * (1080.0 / height), device=device, dtype=torch.float32)
    ph = torch.tensor(rng.uniform(0, 6.28, (3, k)), device=device, dtype=torch.float32)
    amp = torch.tensor(rng.uniform(0.3, 1.0, k), device=device, dtype=torch.float32)
    mats = camera_matrices(n, offset, width, height)
    inv = torch.tensor(np.linalg.inv(mats), device=device, dtype=torch.float32)
    yy, xx = torch.meshgrid(torch.arange(height, device=device, dtype=torch.float32),
                            torch.arange(width, device=device, dtype=torch.float32), indexing="ij")
    out = torch.empty((n, height, width, 3), device=device, dtype=torch.float32)
    norm = float(amp.sum())
    for i in range(n):
        m = inv[i]
        X = m[0, 0] * xx + m[0, 1] * yy + m[0, 2]
        Y = m[1, 0] * xx + m[1, 1] * yy + m[1, 2]
        arg = X[..., None] * fx + Y[..., None] * fy  # [H,W,k]
        for c in range(3):
            v = (torch.sin(arg + ph[c]) * amp).sum(-1) / norm
            out[i, ..., c] = 0.5 + 0.45 * torch.tanh(2.5 * v)
        del arg
    return out


def cpu_baseline(frames_host: np.ndarray, threads: int) -> dict:
    """Time the CPU oracle (reference OpenCV path restated in C) over a bounded sample of the clip."""
    os.environ["OMP_NUM_THREADS"] = str(threads)
    from oracle import oracle as vo

    vo.build()
    import __graft_entry__ as graft

    graft.load_package()
    from vstab_amd import flow_pipeline as fp
    from vstab_amd import host_math as hm

    n, h, w, _ = frames_host.shape
    size = (w, h)
    t0 = time.perf_counter()
    work = hm._working_estimation_size(w, h)
    gray = vo.gray_for_estimation(frames_host, work)
    flow = vo.dis_flow_clip(gray)
    recs = [vo.fit_all_modes(flow[i], 8, "similarity")[0] for i in range(n - 1)]
    work_mats, _, _, _, _ = fp.select_transitions(recs, "similarity")
    mats = [hm._rescale_transform_to_full(m, size, work) if work else m for m in work_mats]
    deltas = np.stack([hm._matrix_to_params(m, "similarity") for m in mats])
    path = np.concatenate([np.zeros((1, 4)), np.cumsum(deltas, axis=0)])
    win = hm.smoothing_window(0.5, 16.0)
    kern = np.ones(win) / win
    sm = np.stack([np.convolve(np.pad(path[:, d], (win // 2,) * 2, mode="edge"), kern, mode="valid") for d in range(4)], 1)
    diffs = 0.7 * (sm - path)
    apply = [hm._params_to_matrix(d, "similarity") for d in diffs]
    mins, maxs = hm._compute_bounding_boxes(apply, w, h)
    x0, y0, x1, y1 = mins[:, 0].max(), mins[:, 1].max(), maxs[:, 0].min(), maxs[:, 1].min()
    shift = np.array([[1, 0, w * 0.5 - (x0 + x1) * 0.5], [0, 1, h * 0.5 - (y0 + y1) * 0.5], [0, 0, 1]], np.float32)
    final = np.stack([shift @ m for m in apply]).astype(np.float32)
    vo.warp_clip(frames_host, final, size, border=hm.border_value((127, 127, 127)))
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "frames/s", "cores": threads, "kind": "port",
            "sample": f"{n} frames of the same synthetic {w}x{h} clip, full path (gray, DIS, fit, trajectory, warp+mask), "
                      f"OpenMP over frames, {dt:.1f} s"}


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--frames", type=int, default=256, help="frames per GPU (BASELINE configs[1]: 256)")
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--force-dist", action="store_true", help="run the sharded/RCCL code path even with one rank (rehearsal)")
    ap.add_argument("--cpu-frames", type=int, default=256, help="frames of the clip timed on the CPU oracle (0 = skip)")
    args = ap.parse_args()

    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and world == 1:
        raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)

    import torch.distributed as dist

    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group(backend="nccl", device_id=device)

    import __graft_entry__ as graft

    graft.load_package()
    from vstab_amd import distributed as vd
    from vstab_amd import flow_pipeline as fp
    from vstab_amd import host_math as hm
    from vstab_amd import native

    ctx = native.Context(local_rank)
    ctx.set_timing(True)

    n_local, h, w = args.frames, args.height, args.width
    total = n_local * world
    start, end = vd.shard_range(total, world, rank)
    halo = 1 if rank > 0 else 0
    frames = synth_clip(n_local + halo, start - halo, h, w, device)
    torch.cuda.synchronize()

    def step():
        if not use_dist:
            context = hm.VideoContext([None] * n_local, hm.FrameAdapter(np.dtype(np.float32), False, "0_1", "torch", False),
                                      w, h, 3, None, "sequence", {}, batch=frames)
            res = fp._stabilize_frames(context, "crop_and_pad", "similarity", False, 0.7, 0.5, 0.6, (127, 127, 127), 16.0,
                                       ctx=ctx, keep_on_device=True)
            return res.frames, res.masks, res.meta
        return vd.stabilize_sharded(ctx, frames, total, "crop_and_pad", "similarity", False, 0.7, 0.5, 0.6, (127, 127, 127), 16.0)

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        out = step()
        del out
    fence()
    ctx.set_timing(True)   # clears the per-kind totals: only the timed steps below are counted
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
        meta = out[2]
        del out
    fence()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = total * args.steps / elapsed
        out_w, out_h = meta["stabilization_warp"]["output_size"]
        # HIP events recorded by the library on the launch stream around every call; summed without host
        # synchronisation inside the timed loop, read here after the closing fence
        stage_ms = {}
        for kind in ("gray", "dis", "fit", "warp"):
            total_ms, launches = ctx.kernel_ms_stats(kind)
            stage_ms[kind] = total_ms / max(launches, 1)
        warp_avg_ms = stage_ms["warp"]
        launch_bytes = WARP_BYTES_PER_PIXEL * out_w * out_h * n_local
        achieved = launch_bytes / (warp_avg_ms * 1e-3) / 1e9
        traffic = None
        tfile = ROOT / "profiles" / "warp_traffic.json"
        if tfile.exists():
            tj = json.loads(tfile.read_text())
            if tj.get("frames") == n_local and tj.get("size") == [w, h]:
                traffic = tj.get("hbm_bytes_per_launch")
        line = {
            "metric": "stabilized frames/sec (1080p, similarity mode)",
            "value": round(value, 2),
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"C2: {n_local}-frame {w}x{h} clip per GPU, Video Stabilizer Flow (DIS) similarity + crop_and_pad, "
                            "defaults (strength 0.7, smooth 0.5, 16 fps), device-resident in/out",
                "frames_per_gpu": n_local,
                "total_frames": total,
                "sharding": "single GPU" if world == 1 else f"contiguous frame shards x{world}, 1-frame halo, RCCL all-gather of fit records",
                "stage_ms": {k: round(float(v), 3) for k, v in stage_ms.items()},
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "warp_kernel<bilinear,q5,mask>",
                "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": traffic,
                "launch_ms": round(warp_avg_ms, 4),
                "algorithmic_bytes_per_launch": launch_bytes,
            },
        }
        if world == 1 and args.cpu_frames >= 2:
            threads = min(16, len(os.sched_getaffinity(0)))
            sample = frames[: min(args.cpu_frames, n_local)].cpu().numpy()
            line["cpu_baseline"] = cpu_baseline(sample, threads)
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
