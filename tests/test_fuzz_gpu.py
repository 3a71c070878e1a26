"""Property tests: drawn shapes, maps and contents through the C ABI against the oracle, bit for bit.  The hand-picked cases of the
other files name the situations somebody thought of; these draw the ones nobody did -- ragged sizes, degenerate maps, borders in
every position.  Deterministic (hypothesis `derandomize`): the same examples on every run."""

import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

pytestmark = pytest.mark.gpu
COMMON = dict(derandomize=True, deadline=None, suppress_health_check=list(HealthCheck), database=None)


def _content(rng, n, h, w):
    kind = rng.integers(0, 4)
    if kind == 0:
        f = rng.random((n, h, w, 3), dtype=np.float32)
    elif kind == 1:                                   # 8-bit-sourced, with flat areas
        f = (rng.integers(0, 256, (n, h, w, 3)) // 64 * 64).astype(np.float32) / np.float32(255.0)
    elif kind == 2:                                   # beyond [0, 1]: bicubic overshoot of an upstream node
        f = rng.uniform(-0.3, 1.4, (n, h, w, 3)).astype(np.float32)
    else:
        yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
        f = np.stack([np.stack([(xx + 3 * i) / max(w, 1), yy / max(h, 1), ((xx + yy) % 5) / 4.0], axis=-1) for i in range(n)]).astype(np.float32)
    return np.ascontiguousarray(f)


def _matrix(rng, sw, sh, dw, dh):
    kind = rng.integers(0, 6)
    th = rng.uniform(-np.pi, np.pi) if kind in (1, 4) else rng.uniform(-0.1, 0.1)
    sc = {0: 1.0, 1: rng.uniform(0.3, 3.0), 2: rng.uniform(0.9, 1.1), 3: rng.uniform(0.02, 0.2), 4: rng.uniform(5.0, 60.0), 5: 1.0}[int(kind)]
    c, s = np.cos(th) * sc, np.sin(th) * sc
    m = np.array([[c, -s, rng.uniform(-0.6, 0.6) * dw], [s, c, rng.uniform(-0.6, 0.6) * dh], [0, 0, 1]], np.float64)
    if kind == 0:
        m[:2, :2] = np.eye(2)                          # pure (sub-pixel) translation
        if rng.integers(0, 2):
            m[0, 2], m[1, 2] = np.round(m[0, 2]), np.round(m[1, 2])
    if kind in (2, 5):
        m[2, 0], m[2, 1] = rng.uniform(-3e-3, 3e-3, 2) if kind == 2 else rng.uniform(-0.2, 0.2, 2) / max(dw, dh)
    return m.astype(np.float32)


@settings(max_examples=250, **COMMON)
@given(st.integers(0, 2 ** 31 - 1), st.integers(1, 3), st.integers(1, 37), st.integers(1, 41), st.integers(1, 45), st.integers(1, 50),
       st.sampled_from(["bilinear", "bicubic"]))
def test_warp_of_drawn_maps_matches_oracle(ctx, oracle, seed, n, sh, sw, dh, dw, interp):
    rng = np.random.default_rng(seed)
    frames = _content(rng, n, sh, sw)
    mats = np.stack([_matrix(rng, sw, sh, dw, dh) for _ in range(n)])
    border = rng.random(3).astype(np.float32)
    ref, ref_mask, ref_cnt = oracle.warp_clip(frames, mats, (dw, dh), interp=interp, border=border)
    dst, mask, cnt = ctx.warp_batch(frames, mats, (dw, dh), interp=interp, border=border, want_count=True)
    assert np.array_equal(dst.cpu().numpy(), ref, equal_nan=True) and np.array_equal(mask.cpu().numpy(), ref_mask)
    assert np.array_equal(cnt.cpu().numpy().astype(np.uint32), ref_cnt)


@settings(max_examples=150, **COMMON)
@given(st.integers(0, 2 ** 31 - 1), st.integers(1, 3), st.integers(2, 70), st.integers(2, 90), st.integers(1, 80), st.integers(1, 100),
       st.sampled_from(["bilinear", "bicubic"]), st.sampled_from([3, 5, 9, 17, 33]), st.floats(0.05, 1.0))
def test_blur_warp_of_drawn_maps_matches_oracle(ctx, oracle, seed, n, sh, sw, dh, dw, interp, samples, blur):
    """warp_blur_kernel picks a path per tile (interior / border-capable staged loop / general loop): whatever it picks, the bits
    are the oracle's."""
    rng = np.random.default_rng(seed)
    frames = _content(rng, n, sh, sw)
    mats = np.stack([_matrix(rng, sw, sh, dw, dh) for _ in range(n)]).astype(np.float64)
    border = rng.random(3).astype(np.float32)
    ref, ref_mask = oracle.warp_blur_clip(frames, mats, (dw, dh), blur, samples, interp=interp, border=border)
    dst, mask = ctx.warp_blur_batch(frames, mats, (dw, dh), blur, samples, interp=interp, border=border)
    assert np.array_equal(dst.cpu().numpy(), ref, equal_nan=True) and np.array_equal(mask.cpu().numpy(), ref_mask, equal_nan=True)


@settings(max_examples=60, **COMMON)
@given(st.integers(0, 2 ** 31 - 1), st.integers(1, 2), st.integers(962, 1500), st.integers(8, 700), st.booleans())
def test_gray_of_drawn_sizes_matches_oracle(ctx, oracle, seed, n, long_side, short_side, portrait):
    """F1 + F2 on sizes nobody picked: the working size from the reference's rounding rule, any area ratio."""
    from vstab_amd import host_math as hm

    rng = np.random.default_rng(seed)
    h, w = (long_side, short_side) if portrait else (short_side, long_side)
    work = hm._working_estimation_size(w, h)          # None where rounding leaves a side unchanged (e.g. 962 x 8): full size
    assert work is None or max(work) == 960
    frames = _content(rng, n, h, w)
    ref = oracle.gray_for_estimation(frames, work)
    got, peaks = ctx.gray_downscale(frames, work, want_range=True)
    assert np.array_equal(got.cpu().numpy(), ref)
    assert np.array_equal(peaks.cpu().numpy(), frames.reshape(n, -1).max(axis=1))


@settings(max_examples=40, **COMMON)
@given(st.integers(0, 2 ** 31 - 1), st.integers(2, 3), st.integers(40, 180), st.integers(40, 260))
def test_dis_of_drawn_sizes_matches_oracle(ctx, oracle, seed, n, h, w):
    """DIS on drawn working sizes (the pyramid depth, the stripe heights and the tile split all follow from them): dense flow and
    the stride-8 grid, bit for bit."""
    import torch

    rng = np.random.default_rng(seed)
    base = rng.integers(0, 256, (h + 24, w + 24)).astype(np.float32)
    for _ in range(2):                               # some structure at several scales
        base = (base + np.roll(base, 1, 0) + np.roll(base, 1, 1) + np.roll(base, (1, 1), (0, 1))) / 4.0
    gray = np.stack([base[12 + (i * 3) % 7: 12 + (i * 3) % 7 + h, 12 + (i * 2) % 5: 12 + (i * 2) % 5 + w] for i in range(n)])
    gray = np.ascontiguousarray(np.clip(gray + rng.normal(0, 2.0, gray.shape), 0, 255).astype(np.uint8))
    ref = oracle.dis_flow_clip(gray)
    flow, grid = ctx.dis_flow_batch(torch.from_numpy(gray), sample_step=8, want_full=True, want_grid=True)
    assert np.array_equal(flow.cpu().numpy(), ref)
    assert np.array_equal(grid.cpu().numpy(), ref[:, ::8, ::8, :])


@settings(max_examples=60, **COMMON)
@given(st.integers(0, 2 ** 31 - 1), st.sampled_from(["translation", "similarity", "perspective"]), st.integers(10, 70), st.integers(10, 125),
       st.floats(0.0, 0.9), st.floats(0.0, 0.5))
def test_fits_of_drawn_flow_fields_match_oracle(ctx, oracle, seed, mode, gh, gw, outliers, holes):
    """F4 + F5 on drawn grids: a camera motion plus noise, a drawn share of gross outliers and of non-finite samples (down to fewer
    than 12 valid points: nothing is computed).  Acceptance, confidence and valid counts equal; matrices as tests/test_fit_gpu.py
    compares them (translation bit-equal; similarity 1e-6; homography 2e-5: its LM solves differ in form)."""
    import torch

    from vstab_amd import native

    rng = np.random.default_rng(seed)
    step = 8
    h, w = gh * step, gw * step
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    th, sc = rng.uniform(-0.02, 0.02), rng.uniform(0.98, 1.02)
    a, b = np.cos(th) * sc - 1.0, np.sin(th) * sc
    flow = np.stack([a * (xx - w / 2) - b * (yy - h / 2) + rng.uniform(-6, 6), b * (xx - w / 2) + a * (yy - h / 2) + rng.uniform(-6, 6)], axis=-1)
    flow = (flow + rng.normal(0, rng.uniform(0.0, 0.6), flow.shape)).astype(np.float32)
    bad = rng.random((h, w)) < outliers
    flow[bad] += rng.uniform(-40, 40, (int(bad.sum()), 2)).astype(np.float32)
    gone = rng.random((h, w)) < holes
    flow[gone] = rng.choice(np.array([np.nan, np.inf, -np.inf], np.float32), (int(gone.sum()), 2))
    grid = np.ascontiguousarray(flow[None, ::step, ::step, :])
    got = native.fit_table_to_dicts(ctx.sample_fit_batch(torch.from_numpy(grid).cuda(), step, mode))[0]
    ref, nv, nt = oracle.fit_all_modes(flow, step, mode)
    assert set(got) == set(ref)
    for name, r in ref.items():
        g = got[name]
        assert g["valid_points"] == nv and g["total_points"] == nt
        assert g["accepted"] == r["accepted"] and g["confidence"] == r["confidence"], name
        if r["accepted"]:
            if name == "translation":
                assert np.array_equal(g["matrix"], r["matrix"])
            elif name == "similarity":
                assert np.allclose(g["matrix"], r["matrix"], rtol=0, atol=1e-6)
            else:
                assert np.allclose(g["matrix"], r["matrix"], rtol=2e-5, atol=1e-7)
            assert g["residual"] == pytest.approx(r["residual"], rel=1e-6)
