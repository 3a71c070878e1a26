"""GPU parity: vstab_dis_flow_batch (HIP) vs oracle/vo_dis.c.

Tolerance: bit-exact.  Both sides use the same operation order (FMA contraction off, correctly
rounded f32 divide/sqrt, OpenCV's 4-lane row-accumulator association for the patch sums -- see
tests/test_dis_sum_order_cpu.py), so every data-dependent branch of the inverse search takes the same path.  The independent check (known synthetic camera
motion) bounds the end result without reference to OpenCV's intermediate values."""

from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]

pytestmark = pytest.mark.gpu


def texture(xx, yy, seed=1234):
    r = np.random.default_rng(seed)
    v = np.zeros_like(xx)
    for _ in range(48):
        fx, fy = r.uniform(-0.25, 0.25, 2)
        ph = r.uniform(0, 6.28)
        a = r.uniform(0.3, 1.0)
        v += a * np.sin(fx * xx + fy * yy + ph)
    return (v - v.min()) / (v.max() - v.min())


def moving_clip(n, h, w, seed=0):
    """u8 gray clip of an analytic texture under a known per-frame similarity (no interpolation)."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    frames, params = [], []
    tx = ty = th = 0.0
    s = 1.0
    for i in range(n):
        if i:
            tx += rng.uniform(-3, 3)
            ty += rng.uniform(-2, 2)
            th += rng.uniform(-0.004, 0.004)
            s *= rng.uniform(0.997, 1.003)
        c, sn = np.cos(th) / s, np.sin(th) / s
        X = c * (xx - w / 2 - tx) + sn * (yy - h / 2 - ty) + w / 2
        Y = -sn * (xx - w / 2 - tx) + c * (yy - h / 2 - ty) + h / 2
        frames.append((texture(X, Y) * 255).astype(np.uint8))
        params.append((tx, ty, th, s))
    return np.stack(frames), params


@pytest.mark.parametrize("n,h,w", [(3, 135, 240), (3, 270, 480), (2, 120, 213), (2, 100, 160),
                                   (4, 45, 73),    # tiny: OpenCV auto-selects finest 0 / coarsest 1
                                   (4, 40, 40),    # tiny + stateful: the first pair uses another pyramid than the rest
                                   (2, 64, 96),
                                   (2, 600, 200),  # portrait: 36 patch rows at the finest level -> 5 rows per stripe (two wavefronts share a stripe)
                                   (2, 960, 540)]) # the tallest working size: 59 patch rows, 8 per stripe
def test_dis_matches_oracle(ctx, oracle, n, h, w):
    import torch

    gray, _ = moving_clip(n, h, w, seed=h)
    ref = oracle.dis_flow_clip(gray)
    flow, grid = ctx.dis_flow_batch(torch.from_numpy(gray), sample_step=8, want_full=True, want_grid=True)
    got = flow.cpu().numpy()
    assert got.shape == ref.shape
    diff = np.abs(got - ref)
    assert np.array_equal(got, ref), f"max abs diff {diff.max()} at {np.unravel_index(diff.argmax(), diff.shape)}"
    assert np.array_equal(grid.cpu().numpy(), ref[:, ::8, ::8, :])


def _content_clips(h, w):
    """Clips that drive the rarely taken branches: zero gradients (determinant floor), pure noise (descent stops at once,
    updates discarded), hard edges, saturated values, a jump far beyond a patch, identical frames."""
    rng = np.random.default_rng(99)
    yy, xx = np.mgrid[0:h, 0:w]
    flat = np.full((3, h, w), 131, np.uint8)
    flat[1] = 7
    noise = rng.integers(0, 256, (3, h, w), dtype=np.uint8)
    bars = np.stack([(((xx + 9 * i) // 11 + (yy - 5 * i) // 17) % 2 * 255).astype(np.uint8) for i in range(3)])
    base, _ = moving_clip(1, h + 80, w + 80, seed=5)
    jump = np.stack([base[0, 40:40 + h, 40:40 + w], base[0, 3:3 + h, 75:75 + w], base[0, 40:40 + h, 40:40 + w]])
    still = np.repeat(base[:, 10:10 + h, 10:10 + w], 3, axis=0)
    sat = np.clip(base[0, :h, :w].astype(np.int32) * 3 - 200, 0, 255).astype(np.uint8)
    sat = np.stack([sat, np.roll(sat, (2, -3), (0, 1)), np.roll(sat, (-1, 4), (0, 1))])
    half = still.copy()
    half[1, :, w // 2:] = noise[1, :, w // 2:]   # one half loses all correspondence
    return {"flat": flat, "noise": noise, "bars": bars, "jump": jump, "still": still, "saturated": sat, "half_noise": half}


@pytest.mark.parametrize("name", ["flat", "noise", "bars", "jump", "still", "saturated", "half_noise"])
@pytest.mark.parametrize("h,w", [(135, 240), (150, 101)])
def test_dis_content_edge_cases_match_oracle(ctx, oracle, name, h, w):
    import torch

    gray = np.ascontiguousarray(_content_clips(h, w)[name])
    ref = oracle.dis_flow_clip(gray)
    flow, _ = ctx.dis_flow_batch(torch.from_numpy(gray), sample_step=8, want_full=True, want_grid=True)
    got = flow.cpu().numpy()
    assert np.isfinite(ref).all()
    diff = np.abs(got - ref)
    assert np.array_equal(got, ref), f"{name}: max abs diff {diff.max()} at {np.unravel_index(diff.argmax(), diff.shape)}"
    if name in ("flat", "still"):
        assert np.abs(got[-1]).max() < (1e-3 if name == "still" else 1.0)


def test_dis_recovers_known_translation(ctx):
    """Independent of the oracle: analytic texture shifted by a known sub-pixel translation."""
    import torch

    h, w = 540, 960
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    f0 = (texture(xx * 0.5, yy * 0.5) * 255).astype(np.uint8)
    f1 = (texture((xx - 5.3) * 0.5, (yy + 2.6) * 0.5) * 255).astype(np.uint8)
    _, grid = ctx.dis_flow_batch(torch.from_numpy(np.stack([f0, f1])), sample_step=8)
    g = grid.cpu().numpy()[0]
    assert g.shape == (68, 120, 2)
    assert abs(np.median(g[..., 0]) - 5.3) < 0.05 and abs(np.median(g[..., 1]) + 2.6) < 0.05


def test_expired_dependency_wait_is_reported(pkg):
    """Fail loudly: when a bounded LDS progress wait of the patch search expires, the kernel records it
    in the context's host-visible status word and the next synchronising call (the fit) returns non-zero with
    vstab_last_error() set -- instead of handing back a wrong flow with rc 0.  The TEST build's VSTAB_DEBUG_PIS_SPIN_LIMIT=0
    makes every not-yet-satisfied wait expire at once (a 540x960 portrait clip, whose stripes have eight patch rows at the
    finest level: two wavefronts share a stripe and wait on each other's rows).  The status word is cleared by the report,
    and the same context computes the same flow as before afterwards.  The shipped library has no such knob: the same
    variable leaves its results unchanged."""
    import os

    import torch
    from tests.util import run_with_hooks
    from vstab_amd import native

    run_with_hooks("""
        from tests.test_dis_gpu import moving_clip
        ctx = native.Context()
        gray, _ = moving_clip(3, 960, 540, seed=7)
        dev = torch.from_numpy(gray).cuda()
        _, clean = ctx.dis_flow_batch(dev, sample_step=8)
        ctx.sample_fit_batch(clean, 8, "similarity")
        clean = clean.clone()
        os.environ["VSTAB_DEBUG_PIS_SPIN_LIMIT"] = "0"
        _, grid = ctx.dis_flow_batch(dev, sample_step=8)
        try:
            ctx.sample_fit_batch(grid, 8, "similarity")
            raise SystemExit("the expired wait was not reported")
        except native.VstabError as exc:
            assert "expired dependency wait" in str(exc), exc
        del os.environ["VSTAB_DEBUG_PIS_SPIN_LIMIT"]
        _, grid = ctx.dis_flow_batch(dev, sample_step=8)
        ctx.sample_fit_batch(grid, 8, "similarity")          # no stale report
        ctx.synchronize()
        assert torch.equal(grid, clean)
        ctx.close()
    """)
    assert native.load_library().vstab_test_hooks() == 0
    ctx = native.Context()
    gray, _ = moving_clip(3, 960, 540, seed=7)
    dev = torch.from_numpy(gray).cuda()
    _, clean = ctx.dis_flow_batch(dev, sample_step=8)
    clean = clean.clone()
    os.environ["VSTAB_DEBUG_PIS_SPIN_LIMIT"] = "0"
    try:
        _, grid = ctx.dis_flow_batch(dev, sample_step=8)
        ctx.sample_fit_batch(grid, 8, "similarity")          # nothing injected, nothing reported
    finally:
        del os.environ["VSTAB_DEBUG_PIS_SPIN_LIMIT"]
    assert torch.equal(grid, clean)
    ctx.close()


@pytest.mark.parametrize("split", ["0", "1", "2"])
@pytest.mark.parametrize("n,h,w", [(3, 270, 480), (4, 45, 73), (2, 540, 960)])
def test_dis_split_and_fused_launch_forms_match_oracle(ctx, oracle, monkeypatch, split, n, h, w):
    """The per-level work runs either as one fused launch (one workgroup per pair) or as split launches that also
    spread over the pixels / tiles inside a pair (few pairs: a multi-GPU rank, a short clip).  Both forms must give
    the oracle's bits; VSTAB_DIS_SPLIT forces the form (the default picks by the number of pairs)."""
    import torch

    monkeypatch.setenv("VSTAB_DIS_SPLIT", split)
    gray, _ = moving_clip(n, h, w, seed=h + 1)
    flow, grid = ctx.dis_flow_batch(torch.from_numpy(gray), sample_step=8, want_full=True, want_grid=True)
    ref = oracle.dis_flow_clip(gray)
    assert np.array_equal(flow.cpu().numpy(), ref)
    assert np.array_equal(grid.cpu().numpy(), ref[:, ::8, ::8, :])


def test_pyramid_tail_launch_equals_the_per_level_launches(ctx, oracle, tmp_path):
    """`pyramid_tail_kernel` (the pyramid's coarse levels + the coarsest level's padded copy / gradients / structure tensor in
    ONE launch) against the per-level launches of rounds 1-4, which VSTAB_DIS_PYRAMID_TAIL=0 brings back (read once per
    process: a child).  Same flow bits, on a clip whose pyramid has general-ratio steps (135 -> 67 -> 33 -> 16 rows) and on
    a small one with fewer levels; both equal the oracle."""
    import subprocess
    import sys

    import torch

    for n, h, w, seed in ((3, 540, 960, 11), (3, 136, 240, 12)):
        gray, _ = moving_clip(n, h, w, seed=seed)
        np.save(tmp_path / "gray.npy", gray)
        code = f"""
import sys; sys.path.insert(0, {str(ROOT)!r})
import numpy as np, torch
import __graft_entry__ as graft
graft.load_package()
from vstab_amd import native
ctx = native.Context(0)
flow, grid = ctx.dis_flow_batch(torch.from_numpy(np.load({str(tmp_path / 'gray.npy')!r})), sample_step=8, want_full=True, want_grid=True)
np.save({str(tmp_path / 'flow_off.npy')!r}, flow.cpu().numpy()); np.save({str(tmp_path / 'grid_off.npy')!r}, grid.cpu().numpy())
"""
        import os

        env = dict(os.environ, VSTAB_DIS_PYRAMID_TAIL="0")
        proc = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        assert proc.returncode == 0, proc.stderr[-3000:]
        flow, grid = ctx.dis_flow_batch(torch.from_numpy(gray), sample_step=8, want_full=True, want_grid=True)
        assert np.array_equal(flow.cpu().numpy(), np.load(tmp_path / "flow_off.npy"))
        assert np.array_equal(grid.cpu().numpy(), np.load(tmp_path / "grid_off.npy"))
        assert np.array_equal(flow.cpu().numpy(), oracle.dis_flow_clip(gray))
