"""world_size-2 gloo tests (CPU) of the sharded Flow path: shard geometry, the all-gather of fit records,
and that every rank derives the same plan as a single process.  The pixel stages need a GPU and are
covered by the -m gpu tests; here the trajectory uses a NumPy stand-in defined in this test."""

import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]


class NumpyTrajectoryCtx:
    """Test double for native.Context.trajectory (flow.py:356-371 + utils.py:361-383 in NumPy)."""

    def trajectory(self, deltas, smooth, fps, strength, camera_lock):
        d = np.asarray(deltas, np.float64)
        n, p = d.shape[0] + 1, d.shape[1]
        path = np.zeros((n, p))
        for i in range(1, n):
            path[i] = path[i - 1] + d[i - 1]
        if camera_lock:
            return path, np.zeros_like(path)
        if smooth <= 0 or n <= 2:
            sm = path.copy()
        else:
            win = max(3, int(round((3 / 16 + smooth * (10 / 16)) * max(1.0, fps))))
            win += 1 - win % 2
            k = np.ones(win) / win
            sm = np.stack([np.convolve(np.pad(path[:, c], (win // 2,) * 2, mode="edge"), k, mode="valid") for c in range(p)], 1)
        return path, path + strength * (sm - path)


def fake_records(total_frames, seed=0):
    """Deterministic per-transition candidate fits (similarity + translation), incl. one forced fallback."""
    rng = np.random.default_rng(seed)
    out = []
    for i in range(total_frames - 1):
        th, s = rng.uniform(-0.01, 0.01), rng.uniform(0.99, 1.01)
        tx, ty = rng.uniform(-3, 3), rng.uniform(-2, 2)
        sim = np.array([[s * np.cos(th), -s * np.sin(th), tx], [s * np.sin(th), s * np.cos(th), ty], [0, 0, 1]], np.float32)
        tr = np.array([[1, 0, tx], [0, 1, ty], [0, 0, 1]], np.float32)
        entry = {
            "similarity": {"matrix": sim, "confidence": 0.9 if i != 9 else 0.05, "residual": 0.2, "accepted": i != 9,
                           "valid_points": 8160, "total_points": 8160},
            "translation": {"matrix": tr, "confidence": 1.0, "residual": 0.4, "accepted": True, "valid_points": 8160,
                            "total_points": 8160},
        }
        out.append({} if i == 4 else entry)
    return out


def _worker(rank, world, port, total_frames, result_dir):
    sys.path.insert(0, str(ROOT))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist

    import __graft_entry__ as graft

    graft.load_package()
    from vstab_amd import distributed as vd
    from vstab_amd import flow_pipeline as fp

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        full = fake_records(total_frames)
        counts = vd.transition_counts(total_frames, world)
        first = sum(counts[:rank])
        local = full[first:first + counts[rank]]
        gathered = vd.gather_fit_records(local, total_frames)
        assert len(gathered) == total_frames - 1
        plan = fp.plan_stabilization(NumpyTrajectoryCtx(), gathered, (192, 108), total_frames, "crop_and_pad", "similarity",
                                     False, 0.7, 0.5, 0.6, (127, 127, 127), 16.0, 16.0)
        start, end = vd.shard_range(total_frames, world, rank)
        import torch

        local_counts = torch.from_numpy((np.arange(start, end, dtype=np.int64) % 5).astype(np.int32))
        fetch = vd._start_gather_counts(local_counts, vd.frame_counts(total_frames, world))   # host rows with gloo
        all_counts, failed_ranks = fetch()
        assert failed_ranks == []
        meta = fp.finish_meta(plan, all_counts)
        np.save(Path(result_dir) / f"final_{rank}.npy", np.stack(plan.final_matrices))
        import json

        (Path(result_dir) / f"meta_{rank}.json").write_text(json.dumps(meta))
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("total_frames", [23, 24])
def test_sharded_plan_matches_single_process(pkg, tmp_path, total_frames):
    import json

    import torch.multiprocessing as mp

    from vstab_amd import flow_pipeline as fp

    world = 2
    mp.spawn(_worker, args=(world, _free_port(), total_frames, str(tmp_path)), nprocs=world, join=True)
    full = fake_records(total_frames)
    plan = fp.plan_stabilization(NumpyTrajectoryCtx(), full, (192, 108), total_frames, "crop_and_pad", "similarity", False, 0.7,
                                 0.5, 0.6, (127, 127, 127), 16.0, 16.0)
    meta = fp.finish_meta(plan, np.arange(total_frames) % 5)
    ref = np.stack(plan.final_matrices)
    for rank in range(world):
        assert np.array_equal(np.load(tmp_path / f"final_{rank}.npy"), ref)
        assert json.loads((tmp_path / f"meta_{rank}.json").read_text()) == json.loads(json.dumps(meta))
    # sticky mode: the forced similarity rejection at transition 9 downgrades everything after it,
    # and the "<12 valid samples" entry at 4 already switched to translation (flow.py:153-154, 338-339)
    modes = [t["mode"] for t in meta["estimated_motion"]["per_transition"]]
    assert modes[:4] == ["similarity"] * 4 and set(modes[4:]) == {"translation"}
    assert meta["transform_mode_applied"] == "translation"


def test_shard_geometry(pkg):
    from vstab_amd import distributed as vd

    for total in (1, 7, 256, 1024, 1030):
        for world in (1, 2, 3, 8):
            spans = [vd.shard_range(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [e - s for s, e in spans]
            assert max(sizes) - min(sizes) <= 1
            if total >= world:
                assert sum(vd.transition_counts(total, world)) == total - 1
    from vstab_amd import native

    recs = fake_records(12)
    back = native.fit_table_to_dicts(native.fit_table_from_dicts(recs))
    for a, b in zip(recs, back):
        assert set(a) == set(b)
        for k in a:
            assert np.array_equal(a[k]["matrix"], b[k]["matrix"]) and a[k]["confidence"] == b[k]["confidence"]
            assert a[k]["accepted"] == b[k]["accepted"]


def _failure_worker(rank, world, port, total_frames, result_dir):
    """Rank 1 fails (a) before the first collective, (b) between the two; both ranks must leave each exchange with the same
    ShardError naming rank 1 -- nobody waits for a peer that has already raised."""
    sys.path.insert(0, str(ROOT))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import datetime
    import json
    import time

    import torch
    import torch.distributed as dist

    import __graft_entry__ as graft

    graft.load_package()
    from vstab_amd import distributed as vd

    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))
    out = {}
    try:
        full = fake_records(total_frames)
        counts = vd.transition_counts(total_frames, world)
        first = sum(counts[:rank])
        local = full[first:first + counts[rank]]
        # (a) the estimation of rank 1 raised: it still joins the exchange, with its status row
        t0 = time.perf_counter()
        try:
            vd.gather_fit_records(None if rank == 1 else local, total_frames, failure=RuntimeError("DIS went wrong on purpose") if rank == 1 else None)
            out["a"] = "no error"
        except vd.ShardError as exc:
            out["a"] = {"failed": exc.failed, "text": str(exc), "s": time.perf_counter() - t0}
        # (a') a rank that hands in the wrong number of transitions is a failure of that rank, reported the same way
        try:
            vd.gather_fit_records(local[:-1] if rank == 1 else local, total_frames)
            out["a2"] = "no error"
        except vd.ShardError as exc:
            out["a2"] = {"failed": exc.failed}
        # the exchange works again afterwards (no half-consumed state)
        assert len(vd.gather_fit_records(local, total_frames)) == total_frames - 1
        # (b) rank 1 failed between the collectives: its status word travels with its (zero) counts
        start, end = vd.shard_range(total_frames, world, rank)
        local_counts = torch.zeros((end - start,), dtype=torch.int32) if rank == 1 else torch.arange(end - start, dtype=torch.int32)
        fetch = vd._start_gather_counts(local_counts, vd.frame_counts(total_frames, world), failed=(rank == 1))
        all_counts, bad = fetch()
        out["b"] = {"bad": bad, "n": int(len(all_counts))}
        fetch = vd._start_gather_counts(local_counts, vd.frame_counts(total_frames, world))
        out["c"] = {"bad": fetch()[1]}
        (Path(result_dir) / f"failure_{rank}.json").write_text(json.dumps(out))
    finally:
        dist.destroy_process_group()


def test_a_failing_rank_is_reported_to_every_rank_through_the_collectives(pkg, tmp_path):
    import json

    import torch.multiprocessing as mp

    world, total = 2, 23
    mp.spawn(_failure_worker, args=(world, _free_port(), total, str(tmp_path)), nprocs=world, join=True)
    for rank in range(world):
        out = json.loads((tmp_path / f"failure_{rank}.json").read_text())
        assert out["a"]["failed"] == [[1, "RuntimeError: DIS went wrong on purpose"]], out
        assert "rank 1" in out["a"]["text"] and out["a"]["s"] < 5.0
        assert [r for r, _ in out["a2"]["failed"]] == [1] and "expected" in out["a2"]["failed"][0][1]
        assert out["b"] == {"bad": [1], "n": total} and out["c"] == {"bad": []}


def test_status_rows_round_trip(pkg):
    from vstab_amd import distributed as vd

    ok = vd._status_row(216, None)
    assert not ok.any() and vd._failed_ranks(np.stack([ok, ok])) == []
    long = ValueError("x" * 1000)
    rows = np.stack([ok, vd._status_row(216, long), vd._status_row(216, KeyError("k"))])
    failed = vd._failed_ranks(rows)
    assert [r for r, _ in failed] == [1, 2] and failed[0][1].startswith("ValueError: xxx") and len(failed[0][1]) == 216 - 4
    assert failed[1][1] == "KeyError: 'k'"
