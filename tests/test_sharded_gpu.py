"""Two ranks (gloo control plane, both on cuda:0 of the one-GPU box) run the sharded Flow pipeline on a
halo-split clip; outputs and meta must equal the single-process result bit for bit."""

import json
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]


def _clip():
    from tests.test_dis_gpu import moving_clip

    gray, _ = moving_clip(9, 270, 480, seed=5)
    f = np.repeat(gray[..., None].astype(np.float32) / 255.0, 3, axis=-1)
    f[..., 2] = 1.0 - f[..., 2]
    return np.ascontiguousarray(f)


def _worker(rank, world, port, out_dir, estimator):
    sys.path.insert(0, str(ROOT))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist

    import __graft_entry__ as graft

    graft.load_package()
    from vstab_amd import distributed as vd
    from vstab_amd import native

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        ctx = native.Context(0)
        frames = _clip()
        n = frames.shape[0]
        start, end = vd.shard_range(n, world, rank)
        halo = 1 if rank > 0 else 0
        local = torch.from_numpy(frames[start - halo:end]).cuda()
        dst, mask, meta = vd.stabilize_sharded(ctx, local, n, "expand", "similarity", False, 0.7, 0.5, 0.6, (127, 127, 127), 16.0,
                                               estimator=estimator)
        np.save(Path(out_dir) / f"dst_{rank}.npy", dst.cpu().numpy())
        np.save(Path(out_dir) / f"mask_{rank}.npy", mask.cpu().numpy())
        (Path(out_dir) / f"meta_{rank}.json").write_text(json.dumps(meta))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("estimator", ["flow", "classic", "flow_phase_correlate"])
def test_two_rank_shards_equal_single_process(pkg, ctx, tmp_path, estimator):
    import torch.multiprocessing as mp

    from vstab_amd import flow_pipeline as fp
    from vstab_amd import host_math as hm

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, str(tmp_path), estimator), nprocs=2, join=True)
    frames = _clip()
    ref = fp._stabilize_frames(hm._normalize_video_input(frames), "expand", "similarity", False, 0.7, 0.5, 0.6, (127, 127, 127), 16.0,
                               estimator=estimator)
    dst = np.concatenate([np.load(tmp_path / f"dst_{r}.npy") for r in range(2)])
    mask = np.concatenate([np.load(tmp_path / f"mask_{r}.npy") for r in range(2)])
    assert np.array_equal(dst, ref.frames) and np.array_equal(mask, ref.masks[..., 0])
    want = json.loads(json.dumps(ref.meta))
    for r in range(2):
        assert json.loads((tmp_path / f"meta_{r}.json").read_text()) == want
