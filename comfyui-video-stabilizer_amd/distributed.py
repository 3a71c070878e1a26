"""Multi-GPU Flow stabilization: frames sharded over ranks, one collective for the motion records.

One process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm, "gloo" in CPU tests).
The reference is single-process (no communication layer); this is the MI355X-native
scale-out of its path, following SURVEY 8(e):

  * rank r owns a contiguous frame range [start_r, end_r) plus ONE halo frame before it
    (the pair at the shard edge needs frame start_r - 1); it runs gray -> DIS -> fit on its own
    frames and produces the candidate fits of transitions start_r-1 -> start_r ... end_r-2 -> end_r-1
  * one all-gather of the per-transition fit records (3 modes x 15 doubles per transition: ~360 B,
    i.e. ~370 KB for a 1024-frame clip) -- latency-bound over xGMI, never bandwidth-bound
  * every rank then replays the same deterministic host logic (sticky mode, prefix sum, box
    filter, framing) on the full record list, so no further exchange is needed before the warp
  * the warp is embarrassingly parallel over the rank's own frames; a second tiny all-gather of the
    per-frame padded-pixel counts completes the meta (padding_fraction_mean/max)
Outputs stay sharded on the devices.
"""

from __future__ import annotations

from typing import Any, Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import host_math as hm
from .flow_pipeline import (_ESTIMATORS, complete_meta, resolve_flow_backend, plan_stabilization,
                            prepare_meta)



def shard_range(total: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous [start, end) of `total` frames owned by `rank`; sizes differ by at most one."""
    base, extra = divmod(int(total), int(world))
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def _gather_rows(local: np.ndarray, counts: Sequence[int], group=None, device=None) -> np.ndarray:
    """all_gather of equally padded row blocks; returns the concatenation of the valid rows in rank order.
    One flat receive buffer and one device->host copy (8 per-rank copies cost more than the collective itself)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rows = max(counts)
    padded = np.zeros((rows,) + local.shape[1:], local.dtype)
    padded[: local.shape[0]] = local
    t = torch.from_numpy(padded)
    if device is not None:
        t = t.to(device)
    flat = torch.empty((world,) + tuple(t.shape), dtype=t.dtype, device=t.device)
    try:
        dist.all_gather_into_tensor(flat, t, group=group)
    except (RuntimeError, NotImplementedError):   # backend without the flat variant
        bucket = [flat[r] for r in range(world)]
        dist.all_gather(bucket, t, group=group)
    host = flat.cpu().numpy()
    return np.concatenate([host[r, : counts[r]] for r in range(world)], axis=0)


def transition_counts(total_frames: int, world: int) -> List[int]:
    """Transitions produced per rank: rank 0 has no halo, so one fewer than its frame count."""
    out = []
    for r in range(world):
        s, e = shard_range(total_frames, world, r)
        out.append((e - s) - (1 if r == 0 else 0))
    return out


def gather_fit_records(local_table, total_frames: int, group=None, device=None):
    """The one data-path collective of the sharded Flow pipeline: all-gather of the raw fit-record table
    (72 B per pair and mode).  Accepts the structured table or the list-of-dicts form."""
    import torch.distributed as dist

    from . import native

    if not isinstance(local_table, np.ndarray):
        local_table = native.fit_table_from_dicts(local_table)
    world = dist.get_world_size(group)
    counts = transition_counts(total_frames, world)
    rank = dist.get_rank(group)
    if local_table.shape[0] != counts[rank]:
        raise ValueError(f"rank {rank} produced {local_table.shape[0]} transitions, expected {counts[rank]}")
    raw = np.ascontiguousarray(local_table).view(np.uint8).reshape(local_table.shape[0], 3 * native.FIT_DTYPE.itemsize)
    full = _gather_rows(raw, counts, group=group, device=device)
    return np.ascontiguousarray(full).view(native.FIT_DTYPE).reshape(-1, 3)


def collective_device(ctx=None):
    """Device tensors must live on for the active backend (RCCL needs HBM tensors, gloo host tensors)."""
    import torch.distributed as dist

    return ctx.device if (ctx is not None and dist.get_backend() == "nccl") else None


def stabilize_sharded(ctx, local_frames, total_frames: int, framing_mode: str, transform_mode: str, camera_lock: bool,
                      strength: float, smooth: float, keep_fov: float, padding_rgb, frame_rate: float, group=None,
                      estimator: str = "flow"):
    """Sharded equivalent of `_stabilize_frames` (flow.py:213-640).

    local_frames: device tensor [n_local (+1 halo for rank > 0), H, W, 3] float32 -- this rank's frames
    preceded by the last frame of the previous rank.  Returns (frames [n_local,h,w,3], masks
    [n_local,h,w], meta) with the outputs resident on this rank's GPU; meta is identical on all ranks."""
    import torch.distributed as dist

    rank, world = dist.get_rank(group), dist.get_world_size(group)
    start, end = shard_range(total_frames, world, rank)
    n_local = end - start
    halo = 1 if rank > 0 else 0
    if local_frames.shape[0] != n_local + halo:
        raise ValueError(f"rank {rank}: got {local_frames.shape[0]} frames, expected {n_local} + {halo} halo")
    height, width = int(local_frames.shape[1]), int(local_frames.shape[2])
    size = (width, height)
    fps_effective = float(max(1.0, frame_rate if (isinstance(frame_rate, (int, float)) and np.isfinite(frame_rate) and frame_rate > 0) else 16.0))
    fps_requested = float(frame_rate) if isinstance(frame_rate, (int, float)) and frame_rate > 0.0 else None
    dev = collective_device(ctx)

    working_size = hm._working_estimation_size(width, height)
    from . import native

    estimator = resolve_flow_backend(estimator)
    estimate = _ESTIMATORS[estimator]
    local_records = (estimate(ctx, local_frames, working_size, transform_mode, clip_start=(rank == 0)) if local_frames.shape[0] >= 2
                     else np.zeros((0, 3), native.FIT_DTYPE))
    records = gather_fit_records(local_records, total_frames, group=group, device=dev)
    plan = plan_stabilization(ctx, records, size, total_frames, framing_mode, transform_mode, camera_lock, strength, smooth,
                              keep_fov, padding_rgb, fps_effective, fps_requested, estimator=estimator)
    if plan.bypass_meta is not None:
        raise NotImplementedError("crop bypass is not wired into the sharded path")
    own = local_frames[halo:]
    mats = np.ascontiguousarray(plan.final_matrices[start:end], dtype=np.float32)
    dst, mask, counts = ctx.warp_batch(own, mats, plan.output_size, interp="bilinear", border=hm.border_value(padding_rgb),
                                       want_mask=True, want_count=True)
    meta = prepare_meta(plan)  # host JSON work overlaps this rank's warp kernel
    frame_counts = [shard_range(total_frames, world, r)[1] - shard_range(total_frames, world, r)[0] for r in range(world)]
    all_counts = _gather_rows(counts.cpu().numpy().astype(np.int64).reshape(-1, 1), frame_counts, group=group, device=dev)
    meta = complete_meta(meta, plan, all_counts.reshape(-1))
    return dst, mask, meta
