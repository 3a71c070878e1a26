"""Smoke self-check used by __graft_entry__.smoke(): one tiny Flow pass and one Motion Apply pass on
cuda:0, compared with the CPU oracle.  This is the only module of the package that touches oracle/
(as the checker; the product path never does)."""

from __future__ import annotations

import sys

import numpy as np


def _moving_clip(n, h, w, seed=0):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    comps = [(rng.uniform(-0.2, 0.2), rng.uniform(-0.2, 0.2), rng.uniform(0, 6.28), rng.uniform(0.3, 1.0)) for _ in range(32)]
    frames = []
    tx = ty = 0.0
    for i in range(n):
        if i:
            tx += rng.uniform(-3, 3)
            ty += rng.uniform(-2, 2)
        v = np.zeros_like(xx)
        for fx, fy, ph, a in comps:
            v += a * np.sin(fx * (xx - tx) + fy * (yy - ty) + ph)
        v = (v - v.min()) / (v.max() - v.min())
        frames.append(np.stack([v, 0.9 * v + 0.05, 1.0 - v], -1).astype(np.float32))
    return np.stack(frames)


def run_smoke(repo_root: str) -> None:
    import torch

    if repo_root not in sys.path:
        sys.path.insert(0, repo_root)
    from oracle import oracle as vo  # checker only

    from . import native

    native.load_library()   # the HIP extension must be there: no CPU fallback exists
    from .nodes import VideoStabilizerFlow, VideoStabilizerMotionApply

    assert torch.cuda.is_available(), "smoke() needs cuda:0"
    torch.cuda.set_device(0)
    frames = _moving_clip(6, 136, 240, seed=1)
    out = VideoStabilizerFlow.execute(torch.from_numpy(frames), 16.0, "crop_and_pad", "similarity", False, 0.7, 0.5, 0.6, "#7F7F7F")
    stab, mask, meta = out[0], out[1], out[2]
    assert tuple(stab.shape) == (6, 136, 240, 3) and tuple(mask.shape) == (6, 136, 240)
    # estimation vs oracle
    gray = vo.gray_for_estimation(frames, None)
    flow = vo.dis_flow_clip(gray)
    for i, tr in enumerate(meta["estimated_motion"]["per_transition"]):
        m, mode, conf, resid = vo.fit_from_flow(flow[i], 8, "similarity")
        assert tr["mode"] == mode and tr["confidence"] == conf, (i, tr["mode"], mode)
        assert np.allclose(np.array(tr["matrix"], np.float32), m, atol=1e-6)
    # warp vs oracle with the matrices the node reports
    mats = np.array([e["applied_matrix"] for e in meta["stabilization_warp"]["per_frame"]], np.float32)
    border = np.array([127, 127, 127], np.float32) / 255.0
    ref, ref_mask, _ = vo.warp_clip(frames, mats, (240, 136), border=border)
    assert np.array_equal(stab.numpy(), ref) and np.array_equal(mask.numpy(), ref_mask)
    # Motion Apply replay with blur
    out2 = VideoStabilizerMotionApply.execute(torch.from_numpy(frames), meta, "crop_and_pad", "bicubic", "#7F7F7F", 0.5, "Draft")
    m64 = np.array([e["matrix"] for e in meta["motion_meta"]["per_frame"]], np.float64)
    ref2, ref2_mask = vo.warp_blur_clip(frames, m64, (240, 136), 0.5, 5, interp="bicubic", border=border)
    assert np.array_equal(out2[0].numpy(), ref2) and np.array_equal(out2[1].numpy(), ref2_mask)
    print("smoke ok: flow + motion-apply on", torch.cuda.get_device_name(0), "match the oracle")
