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

Failure agreement: a rank whose estimation, plan or warp raised still joins both collectives -- with a status row
appended to its fit-record block, and a status word appended to its padded-pixel counts -- and every rank raises the
same `ShardError` once the gathered words are in.  No extra collective, no rank left waiting for a peer that has
already left (soft failures, i.e. an unusable fit of a pair, never raise: video_stabilizer_flow.py:224-225 of the
reference; hard ones do: :275-277).

Motion Apply (`apply_motion_sharded`, BASELINE config C5) needs NO collective: the matrices come from the replicated
motion_meta JSON, `expand` sizes its canvas from the replicated bounding boxes, `crop` ANDs the coverage of all the
clip's matrices on every rank (coverage is a function of the matrices only), and a motion-blurred frame at a shard
edge needs its neighbour's matrix, never its pixels (motion_apply.py:125-134).
"""

from __future__ import annotations

from typing import Any, Dict, List, Optional, Sequence, Tuple

import numpy as np

import os
import time

from . import host_math as hm
from . import flow_pipeline as _fp
from .flow_pipeline import (_ESTIMATORS, _attach_motion_meta, complete_meta, resolve_flow_backend, plan_stabilization,
                            prepare_meta)


class ShardError(RuntimeError):
    """Raised on EVERY rank of a sharded call when any rank failed: `failed` lists (rank, message)."""

    def __init__(self, where: str, failed: Sequence[Tuple[int, str]]):
        self.failed = list(failed)
        detail = "; ".join(f"rank {r}: {m}" for r, m in self.failed)
        super().__init__(f"stabilize_sharded failed {where} on {len(self.failed)} rank(s) -- {detail}")


_STATUS_HEAD = 4   # bytes of a status row before the message: flag, message length (u16 LE), reserved


def _status_row(width: int, failure: Optional[BaseException]) -> np.ndarray:
    """One row of the fit-record exchange that carries this rank's verdict: all zero = fine, else flag + the text."""
    row = np.zeros((width,), np.uint8)
    if failure is not None:
        msg = f"{type(failure).__name__}: {failure}".encode("utf-8", "replace")[: max(0, width - _STATUS_HEAD)]
        row[0] = 1
        row[1], row[2] = len(msg) & 0xFF, len(msg) >> 8
        row[_STATUS_HEAD:_STATUS_HEAD + len(msg)] = np.frombuffer(msg, np.uint8)
    return row


def _failed_ranks(status_rows: np.ndarray) -> List[Tuple[int, str]]:
    out = []
    for r, row in enumerate(np.asarray(status_rows, np.uint8)):
        if row[0]:
            n = int(row[1]) | (int(row[2]) << 8)
            out.append((r, bytes(row[_STATUS_HEAD:_STATUS_HEAD + n]).decode("utf-8", "replace")))
    return out


def _late_failures(bad: Sequence[int], rank: int, own: Optional[BaseException]) -> List[Tuple[int, str]]:
    """The second collective carries one word per rank, no text: a peer's message is in that rank's own exception."""
    return [(r, f"{type(own).__name__}: {own}" if (r == rank and own is not None) else "raised (its own exception has the text)")
            for r in bad]


def _raise_if_failed(where: str, failed: List[Tuple[int, str]], own: Optional[BaseException]):
    if failed:
        raise ShardError(where, failed) from own


def shard_range(total: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous [start, end) of `total` frames owned by `rank`; sizes differ by at most one."""
    base, extra = divmod(int(total), int(world))
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def _gather_rows(local: np.ndarray, counts: Sequence[int], group=None, device=None, failure: Optional[BaseException] = None,
                 with_status: bool = False):
    """all_gather of equally padded row blocks; returns the concatenation of the valid rows in rank order.
    One flat receive buffer and one device->host copy (8 per-rank copies cost more than the collective itself).
    The collective variant is chosen from the backend, never by catching an error: a rank that swallowed a genuine
    RCCL failure and issued a different collective would desynchronise the group.
    with_status (uint8 rows): one more row per rank travels behind its block, the rank's status row (`failure`);
    returns (rows, [(rank, message) of the ranks that failed])."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rows = max(max(counts), 1)
    padded = np.zeros((rows + (1 if with_status else 0),) + local.shape[1:], local.dtype)
    padded[: local.shape[0]] = local
    if with_status:
        padded[rows] = _status_row(padded.shape[1], failure)
    t = torch.from_numpy(padded)
    if device is not None:
        t = t.to(device)
    flat = torch.empty((world,) + tuple(t.shape), dtype=t.dtype, device=t.device)
    if dist.get_backend(group) == "nccl":
        dist.all_gather_into_tensor(flat, t, group=group)
    else:   # gloo (CPU tests): the list form
        dist.all_gather([flat[r] for r in range(world)], t, group=group)
    host = flat.cpu().numpy()
    out = np.concatenate([host[r, : counts[r]] for r in range(world)], axis=0)
    return (out, _failed_ranks(host[:, rows])) if with_status else out


def _start_gather_counts(counts, per_rank: Sequence[int], group=None, on_device: bool = False, failed: bool = False):
    """all-gather of the per-frame padded-pixel counts, in two halves so that host work can run in between.
    With RCCL the int32 device tensor the warp kernel filled goes into the collective as it is, stream-ordered behind the
    kernel (no host round trip before the exchange): this call only ENQUEUES it; the returned function fetches the
    gathered table with one D2H copy.  With gloo (CPU tests) the counts travel as host rows when the function is called.
    One more word per rank travels behind its counts: non-zero = this rank failed between the two collectives (`failed`).
    The returned function gives (counts of all frames, [ranks that failed])."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rows = max(max(per_rank), 1)

    def split(host):
        bad = [r for r in range(world) if int(host[r, rows]) != 0]
        return np.concatenate([host[r, : per_rank[r]] for r in range(world)]).astype(np.int64), bad

    if not on_device:
        def finish_host():
            padded = torch.zeros((rows + 1,), dtype=torch.int32)
            padded[: counts.shape[0]] = counts.cpu()
            padded[rows] = 1 if failed else 0
            parts = [torch.empty_like(padded) for _ in range(world)]
            dist.all_gather(parts, padded, group=group)
            return split(torch.stack(parts).numpy())

        return finish_host
    padded = torch.zeros((rows + 1,), dtype=torch.int32, device=counts.device)
    padded[: counts.shape[0]] = counts
    if failed:
        padded[rows] = 1
    flat = torch.empty((world, rows + 1), dtype=torch.int32, device=counts.device)
    dist.all_gather_into_tensor(flat, padded, group=group)
    return lambda: split(flat.cpu().numpy())


def transition_counts(total_frames: int, world: int) -> List[int]:
    """Transitions produced per rank: rank 0 has no halo, so one fewer than its frame count."""
    out = []
    for r in range(world):
        s, e = shard_range(total_frames, world, r)
        out.append(max(0, (e - s) - (1 if r == 0 else 0)))   # a rank without frames (total < world) has none
    return out


def frame_counts(total_frames: int, world: int) -> List[int]:
    return [shard_range(total_frames, world, r)[1] - shard_range(total_frames, world, r)[0] for r in range(world)]


def gather_fit_records(local_table, total_frames: int, group=None, device=None, failure: Optional[BaseException] = None):
    """The one data-path collective of the sharded Flow pipeline: all-gather of the raw fit-record table
    (72 B per pair and mode) + one status row per rank.  Accepts the structured table or the list-of-dicts form.
    `failure`: this rank's estimation raised (its rows are then zero and ignored) -- every rank, this one included,
    raises ShardError after the exchange."""
    import torch.distributed as dist

    from . import native

    world = dist.get_world_size(group)
    counts = transition_counts(total_frames, world)
    rank = dist.get_rank(group)
    if failure is None:
        try:
            if not isinstance(local_table, np.ndarray):
                local_table = native.fit_table_from_dicts(local_table)
            if local_table.shape[0] != counts[rank]:
                raise ValueError(f"rank {rank} produced {local_table.shape[0]} transitions, expected {counts[rank]}")
        except Exception as exc:   # noqa: BLE001 -- reported through the exchange, raised on every rank below
            failure = exc
    if failure is not None:
        local_table = np.zeros((counts[rank], 3), native.FIT_DTYPE)
    raw = np.ascontiguousarray(local_table).view(np.uint8).reshape(local_table.shape[0], 3 * native.FIT_DTYPE.itemsize)
    full, failed = _gather_rows(raw, counts, group=group, device=device, failure=failure, with_status=True)
    _raise_if_failed("before the exchange of the fit records", failed, failure)
    return np.ascontiguousarray(full).view(native.FIT_DTYPE).reshape(-1, 3)


def collective_device(ctx=None):
    """Device tensors must live on for the active backend (RCCL needs HBM tensors, gloo host tensors)."""
    import torch.distributed as dist

    return ctx.device if (ctx is not None and dist.get_backend() == "nccl") else None


def _lap(stats, key, t0):
    if stats is not None:
        stats[key] = stats.get(key, 0.0) + (time.perf_counter() - t0) * 1e3
    return time.perf_counter()


def stabilize_sharded(ctx, local_frames, total_frames: int, framing_mode: str, transform_mode: str, camera_lock: bool,
                      strength: float, smooth: float, keep_fov: float, padding_rgb, frame_rate: float, group=None,
                      estimator: str = "flow", stats: Optional[Dict[str, float]] = None, want_meta: bool = True,
                      check_value_range: bool = True):
    """Sharded equivalent of `_stabilize_frames` (flow.py:213-640).

    local_frames: device tensor [n_local (+1 halo for rank > 0 that owns frames), H, W, 3] float32 -- this rank's frames
    preceded by the last frame of the previous rank; a rank without frames (total_frames < world) passes [0,H,W,3].
    Returns (frames [n_local,h,w,3], masks [n_local,h,w], meta) with the outputs resident on this rank's GPU; meta is
    identical on all ranks.  `stats` (optional dict) receives host wall-clock milliseconds per phase of this rank:
    estimate (launch + the fit's host sync), gather_fits, plan, warp_launch, meta, gather_counts -- the two gathers and
    plan + meta are the replicated / serial part that bounds strong scaling.
    want_meta=False: this rank skips building the JSON meta (returns None for it) but still joins both collectives --
    a driver that needs the dict once asks rank 0 only.  check_value_range: F0's per-frame `max > 1.5 -> /255` rule
    (stabilizer_utils.py:127-131) is applied to this rank's frames from the maxima the gray pass reports; it is a
    per-frame rule, so no rank needs to know another rank's verdict."""
    import torch.distributed as dist

    from . import native

    rank, world = dist.get_rank(group), dist.get_world_size(group)
    if total_frames < 2:
        # flow.py:242-310 (empty / single-frame passthrough) has nothing to shard; every rank raises alike, before
        # any collective
        raise ValueError(f"stabilize_sharded needs a clip of at least 2 frames, got {total_frames}")
    start, end = shard_range(total_frames, world, rank)
    n_local = end - start
    halo = 1 if (rank > 0 and n_local > 0) else 0
    if local_frames.shape[0] != n_local + halo:
        raise ValueError(f"rank {rank}: got {local_frames.shape[0]} frames, expected {n_local} + {halo} halo")
    height, width = int(local_frames.shape[1]), int(local_frames.shape[2])
    size = (width, height)
    fps_effective = float(max(1.0, frame_rate if (isinstance(frame_rate, (int, float)) and np.isfinite(frame_rate) and frame_rate > 0) else 16.0))
    fps_requested = float(frame_rate) if isinstance(frame_rate, (int, float)) and frame_rate > 0.0 else None
    dev = collective_device(ctx)
    torch = ctx.torch

    working_size = hm._working_estimation_size(width, height)
    estimator = resolve_flow_backend(estimator)
    estimate = _ESTIMATORS[estimator]
    t0 = time.perf_counter()
    # A function of the call's arguments and the backend only, hence the same on every rank: the ranks issue the same two
    # collectives either way (the fit records' all-gather has one shape in both forms).
    # VSTAB_SHARDED_DEVICE_PLAN=force: take the device-plan form under a gloo control plane too (its collectives then go
    # through the host) -- how the multi-rank layout of that form is tested with several ranks on a one-GPU box.
    forced = os.environ.get("VSTAB_SHARDED_DEVICE_PLAN", "") == "force"
    if (dev is not None or forced) and _fp.device_plan_applies(estimator, framing_mode, transform_mode, total_frames,
                                                                            segments=world):
        return _stabilize_sharded_device_plan(ctx, local_frames, total_frames, start, n_local, halo, size, working_size,
                                              framing_mode, transform_mode, camera_lock, strength, smooth, keep_fov, padding_rgb,
                                              fps_effective, fps_requested, group, stats, want_meta, check_value_range, t0)
    # Every rank issues exactly two collectives, whatever happens to it in between: an exception is parked, reported
    # through the collective that follows, and raised -- on every rank -- once the gathered status words are in.
    failure: Optional[BaseException] = None
    local_records = None
    try:
        if local_frames.shape[0] >= 2:
            peaks = [] if check_value_range else None
            local_records = estimate(ctx, local_frames, working_size, transform_mode, clip_start=(rank == 0), peaks_out=peaks)
            if peaks:
                rescaled, _ = hm.apply_value_range(local_frames, peaks[0], ctx)
                if rescaled is not local_frames:   # 0..255 float frames on this rank: estimate again on the rescaled ones
                    local_frames = rescaled
                    local_records = estimate(ctx, local_frames, working_size, transform_mode, clip_start=(rank == 0))
        else:
            if check_value_range and local_frames.shape[0] == 1:
                local_frames, _ = hm.apply_value_range(local_frames, ctx.frame_range(local_frames), ctx)
            local_records = np.zeros((0, 3), native.FIT_DTYPE)
    except Exception as exc:   # noqa: BLE001 -- travels in this rank's status row
        failure = exc
    t0 = _lap(stats, "estimate", t0)
    records = gather_fit_records(local_records, total_frames, group=group, device=dev, failure=failure)
    t0 = _lap(stats, "gather_fits", t0)
    own = local_frames[halo:]
    plan = meta = dst = mask = None
    counts = torch.zeros((0,), dtype=torch.int32, device=own.device)
    try:
        plan = plan_stabilization(ctx, records, size, total_frames, framing_mode, transform_mode, camera_lock, strength, smooth,
                                  keep_fov, padding_rgb, fps_effective, fps_requested, estimator=estimator)
        t0 = _lap(stats, "plan", t0)
        if plan.bypass_meta is not None:   # crop + keep_fov ~ 1 (flow.py:387-429): the original frames, zero masks
            # (the decision is replicated -- a pure function of the gathered records and the call's arguments -- but the
            # second collective is issued all the same: its status word is what tells the others if THIS rank failed)
            dst, mask = own, torch.zeros((n_local, height, width), dtype=torch.float32, device=own.device)
        elif n_local > 0:
            mats = np.ascontiguousarray(plan.final_matrices[start:end], dtype=np.float32)
            dst, mask, counts = ctx.warp_batch(own, mats, plan.output_size, interp="bilinear", border=hm.border_value(padding_rgb),
                                               want_mask=True, want_count=True)
        else:   # nothing to warp here, but this rank still takes part in the second collective below
            out_w, out_h = plan.output_size
            dst = torch.empty((0, out_h, out_w, 3), dtype=torch.float32, device=own.device)
            mask = torch.empty((0, out_h, out_w), dtype=torch.float32, device=own.device)
    except Exception as exc:   # noqa: BLE001 -- travels in this rank's status word
        failure = exc
    bypass = plan is not None and plan.bypass_meta is not None
    per_frame = frame_counts(total_frames, world)   # the collective's shape: replicated information only
    if len(counts) != per_frame[rank]:              # bypass, or this rank failed before its warp: zeros travel
        counts = torch.zeros((per_frame[rank],), dtype=torch.int32, device=own.device)
    fetch_counts = _start_gather_counts(counts, per_frame, group=group, on_device=dev is not None, failed=failure is not None)
    t0 = _lap(stats, "warp_launch", t0)
    late: Optional[BaseException] = None
    try:   # host JSON work overlaps this rank's warp kernel and the collective
        if failure is None and want_meta and not bypass:
            meta = prepare_meta(plan)
    except Exception as exc:   # noqa: BLE001 -- after the status word left: raised here once the collective is joined
        late = exc
    t0 = _lap(stats, "meta", t0)
    all_counts, bad = fetch_counts()
    _raise_if_failed("between its two collectives", _late_failures(bad, rank, failure), failure)
    if late is not None:
        raise late
    if bypass:
        return dst, mask, _attach_motion_meta(plan.bypass_meta, fps_effective, estimator)
    if want_meta:
        meta = complete_meta(meta, plan, all_counts)
    _lap(stats, "gather_counts", t0)
    if stats is not None:
        stats["device_plan"] = {"used": False, "mismatched_frames": 0}
    return dst, mask, meta


def _stabilize_sharded_device_plan(ctx, local_frames, total_frames, start, n_local, halo, size, working_size, framing_mode,
                                   transform_mode, camera_lock, strength, smooth, keep_fov, padding_rgb, fps_effective,
                                   fps_requested, group, stats, want_meta, check_value_range, t0):
    """stabilize_sharded with the plan formed on the device (flow_pipeline: "the plan formed on the device, speculatively"),
    RCCL only: a rank's fit records go from the fit kernel into the all-gather's send buffer on the device, the gathered
    table feeds plan_kernel where RCCL leaves it, and the rank's warp is queued behind that -- the rank's GPU never waits
    for its host.  Every rank then downloads the gathered table, forms the replicated host plan (exact, the reference's
    arithmetic) WHILE its warp runs, and checks its own frames' matrices against the device plan's."""
    import torch.distributed as dist

    from . import native

    torch = ctx.torch
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    rec_bytes = 3 * native.FIT_DTYPE.itemsize
    per_rank = transition_counts(total_frames, world)
    rows = max(max(per_rank), 1)
    send = torch.zeros((rows + 1, rec_bytes), dtype=torch.uint8, device=ctx.device)   # + this rank's status row
    pairs_local = 0
    failure: Optional[BaseException] = None   # parked, reported through the next collective, raised on every rank (see stabilize_sharded)
    try:
        if local_frames.shape[0] >= 2:
            def launch(frames):
                peaks_ = [] if check_value_range else None
                gray = _fp._gray(ctx, frames, working_size, peaks_)
                _, grid = ctx.dis_flow_batch(gray, sample_step=_fp.SAMPLE_STEP, want_full=False, want_grid=True, clip_start=(rank == 0))
                return ctx.sample_fit_batch_begin(grid, _fp.SAMPLE_STEP, transform_mode), peaks_

            pairs_local, peaks = launch(local_frames)
            if peaks:
                rescaled, _ = hm.apply_value_range(local_frames, peaks[0], ctx)
                if rescaled is not local_frames:   # 0..255 float frames on this rank: estimate again on the rescaled ones
                    ctx.sample_fit_batch_end(pairs_local)
                    local_frames = rescaled
                    check_value_range = False
                    pairs_local, _ = launch(local_frames)
            ctx.fit_records_copy(send, pairs_local)
        elif check_value_range and local_frames.shape[0] == 1:
            local_frames, _ = hm.apply_value_range(local_frames, ctx.frame_range(local_frames), ctx)
        if pairs_local != per_rank[rank]:
            raise ValueError(f"rank {rank} produced {pairs_local} transitions, expected {per_rank[rank]}")
    except Exception as exc:   # noqa: BLE001 -- travels in this rank's status row
        failure = exc
        send[rows].copy_(torch.from_numpy(_status_row(rec_bytes, exc)))
    t0 = _lap(stats, "estimate", t0)
    flat = torch.empty((world, rows + 1, rec_bytes), dtype=torch.uint8, device=ctx.device)
    on_device = dist.get_backend(group) == "nccl"
    if on_device:
        dist.all_gather_into_tensor(flat, send, group=group)
    else:   # gloo (tests of this form on a one-GPU box): the same bytes into the same places, through the host
        parts = [torch.empty((rows + 1, rec_bytes), dtype=torch.uint8) for _ in range(world)]
        dist.all_gather(parts, send.cpu(), group=group)
        flat.copy_(torch.stack(parts))
    # the host's copy of the gathered table is queued HERE, ahead of the plan kernel and the warp in stream order, into
    # page-locked memory with an event of its own: a plain .cpu() after the launches would wait for the warp to finish
    host_t = torch.empty(flat.shape, dtype=torch.uint8, pin_memory=True)
    host_t.copy_(flat, non_blocking=True)
    gathered = torch.cuda.Event()
    gathered.record()
    own = local_frames[halo:]
    out_size = size
    dst = torch.empty((0, size[1], size[0], 3), dtype=torch.float32, device=own.device)
    mask = torch.empty((0, size[1], size[0]), dtype=torch.float32, device=own.device)
    counts = torch.zeros((0,), dtype=torch.int32, device=own.device)
    late: Optional[BaseException] = None      # what happens to this rank between the collectives
    if failure is None:
        try:
            # (a peer that failed sent zero records: plan_kernel treats such pairs as "no fit" -- identity -- so the launches
            # below are harmless; this rank learns of the failure from the status rows a few lines down and raises)
            ctx.flow_plan_device(flat.data_ptr(), total_frames - 1, transform_mode, size, working_size, smooth, fps_effective, strength,
                                 bool(camera_lock), seg_pairs=per_rank, seg_rows=rows + 1, warp_frames=n_local, framing=framing_mode)
            if framing_mode == "expand":   # the canvas comes from the (replicated) plan kernel's region, see flow_pipeline
                out_size = ctx.expand_canvas(ctx.flow_plan_result(total_frames, _fp.PLAN_PARAMS[transform_mode])[3])
                if out_size is None:
                    raise native.VstabError("the device plan's expand region is not finite")
                dst = torch.empty((0, out_size[1], out_size[0], 3), dtype=torch.float32, device=own.device)
                mask = torch.empty((0, out_size[1], out_size[0]), dtype=torch.float32, device=own.device)
            if n_local > 0:
                dst, mask, counts = ctx.warp_batch_planned(own, start, out_size, border=hm.border_value(padding_rgb), want_mask=True,
                                                           want_count=True)
            t0 = _lap(stats, "warp_launch", t0)
            if pairs_local:
                ctx.sample_fit_batch_end(pairs_local)      # this rank's fits are done (and its DIS reported no failure)
        except Exception as exc:   # noqa: BLE001 -- travels in this rank's status word of the second collective
            late = exc
    gathered.synchronize()                         # waits for the all-gather and its download, not for the warp
    host = host_t.numpy()
    _raise_if_failed("before the exchange of the fit records", _failed_ranks(host[:, rows]), failure)
    plan = meta = None
    verdict = {"used": True, "mismatched_frames": 0}
    if late is None:
        try:
            records = np.ascontiguousarray(np.concatenate([host[r, : per_rank[r]] for r in range(world)], axis=0)).view(native.FIT_DTYPE).reshape(-1, 3)
            t0 = _lap(stats, "gather_fits", t0)
            plan = plan_stabilization(ctx, records, size, total_frames, framing_mode, transform_mode, camera_lock, strength, smooth,
                                      keep_fov, padding_rgb, fps_effective, fps_requested, estimator="flow")
            t0 = _lap(stats, "plan", t0)
            final_dev = ctx.flow_plan_result(total_frames, _fp.PLAN_PARAMS[transform_mode])[0]
            if tuple(plan.output_size) != tuple(out_size):
                # expand: the host's canvas differs from the device's by a pixel (an extent within one ulp of an integer) -- on every
                # rank alike, the plan being replicated: this rank's frames are warped again onto the host plan's canvas
                if n_local > 0:
                    dst, mask, counts = ctx.warp_batch(own, np.ascontiguousarray(plan.final_matrices[start:start + n_local], np.float32),
                                                       plan.output_size, interp="bilinear", border=hm.border_value(padding_rgb),
                                                       want_mask=True, want_count=True)
                else:
                    dst = torch.empty((0, plan.output_size[1], plan.output_size[0], 3), dtype=torch.float32, device=own.device)
                    mask = torch.empty((0, plan.output_size[1], plan.output_size[0]), dtype=torch.float32, device=own.device)
                verdict["mismatched_frames"] = n_local
            elif n_local > 0:
                sub = _fp.FlowPlan(plan.final_matrices[start:start + n_local], plan.output_size, {}, {}, {}, plan.framing_mode, size, fps_effective)
                verdict["mismatched_frames"] = _fp._rewarp_mismatched(ctx, own, sub, final_dev[start:start + n_local], dst, mask, counts,
                                                                      padding_rgb)
        except Exception as exc:   # noqa: BLE001
            late = exc
    per_frame = frame_counts(total_frames, world)
    if late is not None and len(counts) != per_frame[rank]:
        counts = torch.zeros((per_frame[rank],), dtype=torch.int32, device=own.device)
    fetch_counts = _start_gather_counts(counts, per_frame, group=group, on_device=on_device, failed=late is not None)
    after: Optional[BaseException] = None
    try:   # host JSON work overlaps this rank's warp kernel and the collective
        if late is None and want_meta:
            meta = prepare_meta(plan)
    except Exception as exc:   # noqa: BLE001 -- after the status word left: raised here once the collective is joined
        after = exc
    t0 = _lap(stats, "meta", t0)
    all_counts, bad = fetch_counts()
    _raise_if_failed("between its two collectives", _late_failures(bad, rank, late), late)
    if after is not None:
        raise after
    if want_meta:
        meta = complete_meta(meta, plan, all_counts)
    _lap(stats, "gather_counts", t0)
    if stats is not None:
        stats["device_plan"] = verdict
    return dst, mask, meta


def apply_motion_sharded(ctx, local_frames, start: int, total_frames: int, meta: Dict[str, Any], padding_rgb, *,
                         framing_mode: str = "crop_and_pad", interpolation: str = "bilinear", motion_blur: float = 0.0,
                         motion_blur_samples: int = 9, progress_callback=None):
    """Sharded `apply_motion` (motion_apply.py:297-429): this rank replays frames [start, start + n_local) of a clip of
    `total_frames` frames whose motion is described by the replicated `meta`.  No collective, no halo frame.

    local_frames: device tensor [n_local, H, W, 3] float32 (may be empty).  Returns (frames [n_local,h,w,3], masks
    [n_local,h,w], result_meta) on this rank's GPU; result_meta is identical on all ranks.  The validation errors of
    the reference (size mismatch, frame-count mismatch against the metadata, bad enums) are raised on every rank alike."""
    from .apply_pipeline import _resolve_motion_for_context, _validate_context, apply_motion_on_device

    n_local = int(local_frames.shape[0])
    height, width = int(local_frames.shape[1]), int(local_frames.shape[2])
    if start < 0 or start + n_local > total_frames:
        raise ValueError(f"shard [{start}, {start + n_local}) lies outside a clip of {total_frames} frames")
    clip = hm.VideoContext([None] * int(total_frames), hm.FrameAdapter(np.dtype(np.float32), False, "0_1", "torch", False),
                           width, height, 3, None, "sequence", {})
    motion = _resolve_motion_for_context(meta, clip)
    _validate_context(clip, motion)
    return apply_motion_on_device(ctx, local_frames, int(start), motion, meta, padding_rgb, framing_mode=framing_mode,
                                  interpolation=interpolation, motion_blur=motion_blur,
                                  motion_blur_samples=motion_blur_samples, progress_callback=progress_callback)
