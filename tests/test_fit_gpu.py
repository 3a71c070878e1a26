"""GPU parity: vstab_sample_fit_batch / vstab_trajectory vs the oracle.

Tolerances (floating point, stated per SURVEY 8c):
  * RANSAC inlier counts -> confidence: exact (integer counts, same RNG sequence, same f32 error test)
  * similarity matrix: 1e-6 abs (GPU solves the least-squares normal equations in closed form, the oracle
    runs OpenCV's 10 LM iterations; both converge to the same minimiser, f32 rounding of the result may
    differ in the last bit)
  * homography: 2e-5 relative on the entries (same RANSAC samples and inlier set; the LM steps solve the
    8x8 normal equations by elimination instead of OpenCV's eigen-solve, sums in a different order)
  * translation (median): exact
  * residuals: 1e-6 relative (fp64 sums in a different order)
"""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def synth_flow(h, w, kind, seed):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    th, s = rng.uniform(-0.01, 0.01), rng.uniform(0.99, 1.01)
    tx, ty = rng.uniform(-6, 6), rng.uniform(-4, 4)
    c, sn = s * np.cos(th), s * np.sin(th)
    g0, g1 = rng.uniform(-2e-5, 2e-5, 2)
    den = g0 * xx + g1 * yy + 1.0
    fx = (c * xx - sn * yy + tx) / den - xx
    fy = (sn * xx + c * yy + ty) / den - yy
    flow = np.stack([fx, fy], -1).astype(np.float32)
    flow += rng.normal(0, 0.15, flow.shape).astype(np.float32)
    if kind == "outliers":
        m = rng.random((h, w)) < 0.35
        flow[m] += rng.uniform(-25, 25, (int(m.sum()), 2)).astype(np.float32)
    elif kind == "garbage":
        flow = rng.uniform(-40, 40, flow.shape).astype(np.float32)
    elif kind == "nonfinite":
        flow[::16, ::24] = np.nan
        flow[8::32, 8::16, 0] = np.inf
    return flow


@pytest.mark.parametrize("kind", ["clean", "outliers", "garbage", "nonfinite"])
@pytest.mark.parametrize("mode", ["similarity", "translation", "perspective"])
def test_fit_matches_oracle(ctx, oracle, kind, mode):
    import torch

    h, w, step = 135 * 2, 240 * 2, 8
    flows = np.stack([synth_flow(h, w, kind, seed) for seed in range(4)])
    grid = np.ascontiguousarray(flows[:, ::step, ::step, :])
    from vstab_amd import native

    got = native.fit_table_to_dicts(ctx.sample_fit_batch(torch.from_numpy(grid).cuda(), step, mode))
    for p in range(flows.shape[0]):
        ref, nv, nt = oracle.fit_all_modes(flows[p], step, mode)
        assert set(got[p]) == set(ref)
        for name, r in ref.items():
            g = got[p][name]
            assert g["valid_points"] == nv and g["total_points"] == nt
            assert g["accepted"] == r["accepted"], (kind, name)
            assert g["confidence"] == r["confidence"], (kind, name)
            if r["accepted"]:
                if name == "translation":
                    assert np.array_equal(g["matrix"], r["matrix"])
                elif name == "similarity":
                    assert np.allclose(g["matrix"], r["matrix"], rtol=0, atol=1e-6)
                else:  # homography: LM on the GPU solves its 8x8 systems by elimination, the oracle by eigen-solve
                    assert np.allclose(g["matrix"], r["matrix"], rtol=2e-5, atol=1e-7)
                assert g["residual"] == pytest.approx(r["residual"], rel=1e-6)


def test_fit_too_few_valid_points(ctx):
    """flow.py:153-154: fewer than 12 finite samples -> nothing is computed (host falls back to identity)."""
    import torch

    grid = np.full((1, 6, 8, 2), np.nan, np.float32)
    grid[0, 0, :5] = 1.0
    from vstab_amd import native

    got = native.fit_table_to_dicts(ctx.sample_fit_batch(torch.from_numpy(grid).cuda(), 8, "similarity"))
    assert got[0] == {}


@pytest.mark.parametrize("smooth,fps", [(0.0, 16.0), (0.5, 16.0), (1.0, 24.0), (0.5, 60.0), (0.3, 30.0)])
@pytest.mark.parametrize("p,n", [(2, 97), (4, 97), (8, 97), (4, 2048), (8, 2048), (8, 5000)])
def test_trajectory_matches_numpy(ctx, smooth, fps, p, n):
    """fp64 restatement of flow.py:356-371 + utils.py:361-383 written with NumPy in the test, against `vstab_trajectory`
    -- HOST arithmetic since round 4 (csrc/vstab_traj.hip: the same operation order as plan_kernel, no launch; the device
    form is tied to it by tests/test_device_plan_gpu.py).  The sizes once selected between an LDS and a global-memory
    kernel; they stay as short / long / wide paths (up to 8 x 5000 doubles)."""
    rng = np.random.default_rng(int(fps) + p)
    deltas = rng.normal(0, 1.5, (n - 1, p))
    path, target = ctx.trajectory(deltas, smooth, fps, 0.7, False)
    ref_path = np.zeros((n, p))
    for i in range(1, n):
        ref_path[i] = ref_path[i - 1] + deltas[i - 1]
    assert np.array_equal(path, ref_path)
    if smooth <= 0:
        sm = ref_path.copy()
    else:
        window = int(round((3.0 / 16.0 + smooth * (13.0 / 16.0 - 3.0 / 16.0)) * max(1.0, fps)))
        window = max(3, window)
        window += 1 - window % 2
        k = np.ones(window) / window
        sm = np.stack([np.convolve(np.pad(ref_path[:, d], (window // 2,) * 2, mode="edge"), k, mode="valid") for d in range(p)], 1)
    ref_target = ref_path + 0.7 * (sm - ref_path)
    assert np.allclose(target, ref_target, rtol=0, atol=1e-10)
    _, locked = ctx.trajectory(deltas, smooth, fps, 0.7, True)
    assert np.all(locked == 0.0)


def test_trajectory_matches_reference_smooth_golden(ctx):
    """F8 pinned: the shipped `vstab_trajectory` (the host form every plan the node returns is made with) against the 60
    outputs of the REFERENCE's own `_smooth_path` (nodes/stabilizer_utils.py:361-383;
    tests/golden/reference_helpers.json["smooth"], generated by make_golden.py:95-99 from the imported reference).
    vstab_trajectory takes per-frame deltas and starts its path at 0, so each golden path is fed as np.diff and its first
    row is added back; with strength 1 the target IS the smoothed path (flow.py:368: path + 1*(smooth - path)).
    Tolerance 1e-12 absolute on values of magnitude <= 30: the golden is np.convolve's summation order, the library's a
    sequential window sum (observed difference <= 1e-14).  The speculative device form (plan_kernel) holds this pin only
    through its tie to the host form: bit-equal on translation deltas, within 16 ulp on similarity deltas
    (tests/test_device_plan_gpu.py::test_host_trajectory_*)."""
    import json
    from pathlib import Path

    gold = json.loads((Path(__file__).parent / "golden" / "reference_helpers.json").read_text())
    paths = {k: np.array(v["__nd__"], dtype=np.float64) for k, v in gold["smooth_paths"].items()}
    assert len(gold["smooth"]) == 60
    worst = 0.0
    for case in gold["smooth"]:
        path = paths[case["path"]]
        want = np.array(case["out"]["__nd__"], dtype=np.float64)
        got_path, target = ctx.trajectory(np.diff(path, axis=0), case["smooth"], case["fps"], 1.0, False)
        assert np.allclose(got_path + path[0], path, rtol=0, atol=1e-12)
        err = float(np.max(np.abs(target + path[0] - want)))
        worst = max(worst, err)
        assert err <= 1e-12, (case["path"], case["smooth"], case["fps"], err)
    assert worst <= 1e-12
