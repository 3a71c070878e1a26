"""The real-OpenCV tier's seam (oracle/cv2_tier.py, bench.cv2_leg): live the first time a `cv2` imports, tested today.

No cv2 exists in the build container or on the GPU box.  What can be tested without one:
  * the probe says "absent" and the bench keeps kind "port";
  * under the oracle-backed `cv2` stand-in (tests/golden/cv2_standin.py, which accepts only the call forms the reference
    uses and raises on anything else) the tier's calls go through and return exactly what the oracle returns -- i.e. the
    seam's plumbing (argument forms, array layouts, record format, mask arithmetic, blur accumulation) is right, so a real
    cv2 would be compared on equal terms; the stand-in itself is never accepted as "opencv" by the probe.
The last test is the one that matters the day a real cv2 is reachable: oracle vs OpenCV, primitive by primitive."""

import sys

import numpy as np
import pytest

from tests.util import synth_frames, test_matrices as make_matrices


@pytest.fixture()
def standin(oracle):
    from tests.golden import cv2_standin

    saved = sys.modules.get("cv2")
    cv2_standin.install()
    yield cv2_standin
    if saved is None:
        sys.modules.pop("cv2", None)
    else:
        sys.modules["cv2"] = saved


def _real_cv2():
    from oracle import cv2_tier

    return cv2_tier.available(allow_standin=False)


def test_probe_states_absent_when_no_cv2_imports():
    from oracle import cv2_tier

    if _real_cv2():
        pytest.skip("a real cv2 is importable here")
    assert cv2_tier.probe() == {"cv2": "absent"} and not cv2_tier.available()


def test_standin_is_never_taken_for_opencv(standin):
    from oracle import cv2_tier

    info = cv2_tier.probe()
    assert info["cv2"].startswith("oracle-standin") and info["standin"] is True
    assert not cv2_tier.available() and cv2_tier.available(allow_standin=True)


def test_tier_calls_have_the_reference_forms_and_equal_the_oracle_under_the_standin(standin, oracle):
    from oracle import cv2_tier

    tier = cv2_tier.Cv2Tier()
    frames = synth_frames(4, 270, 480, seed=3)
    frames[1:] = np.stack([np.roll(frames[0], (k, 2 * k), axis=(0, 1)) for k in (1, 2, 3)])
    assert np.array_equal(tier.frame_max(frames), oracle.frame_max(frames))
    for work in (None, (240, 135)):
        assert np.array_equal(tier.gray_for_estimation(frames, work), oracle.gray_for_estimation(frames, work))
    gray = oracle.gray_for_estimation(frames, None)
    flow = tier.dis_flow_clip(gray)
    assert np.array_equal(flow, oracle.dis_flow_clip(gray))
    for mode in ("translation", "similarity", "perspective"):
        a, nva, nta = tier.fit_all_modes(flow[0], 8, mode)
        b, nvb, ntb = oracle.fit_all_modes(flow[0], 8, mode)
        assert (nva, nta) == (nvb, ntb) and a.keys() == b.keys()
        for k in a:
            assert a[k]["accepted"] == b[k]["accepted"] and a[k]["confidence"] == b[k]["confidence"]
            assert np.array_equal(a[k]["matrix"], b[k]["matrix"])
            assert abs(a[k]["residual"] - b[k]["residual"]) <= 1e-6 * max(1.0, abs(b[k]["residual"]))   # f32 vs f64 mean
    # too few finite samples: no candidates at all (flow.py:153-154)
    bad = flow[0].copy()
    bad[:] = np.nan
    assert tier.fit_all_modes(bad, 8, "similarity")[0] == {} and oracle.fit_all_modes(bad, 8, "similarity")[0] == {}
    mats = make_matrices(4, 480, 270, "similarity", seed=5)
    border = (0.25, 0.5, 0.75)
    for interp in ("bilinear", "bicubic"):
        d1, m1, c1 = tier.warp_clip(frames, mats, (500, 280), interp=interp, border=border)
        d2, m2, c2 = oracle.warp_clip(frames, mats, (500, 280), interp=interp, border=border)
        assert np.array_equal(d1, d2) and np.array_equal(m1, m2) and np.array_equal(c1, c2)
        b1, k1 = tier.warp_blur_clip(frames, mats, (480, 270), 0.5, 5, interp=interp, border=border)
        b2, k2 = oracle.warp_blur_clip(frames, mats, (480, 270), 0.5, 5, interp=interp, border=border)
        assert np.array_equal(b1, b2) and np.array_equal(k1, k2)


def test_cpu_baseline_runs_either_provider_through_the_same_plan(standin, oracle, pkg):
    import bench
    from oracle import cv2_tier

    frames = np.stack([np.roll(synth_frames(1, 270, 480, seed=9)[0], (k, -k), axis=(0, 1)) for k in range(5)])
    port_line, port = bench.cpu_baseline(frames, 2, keep_outputs=True)
    cv_line, cv = bench.cpu_baseline(frames, 2, keep_outputs=True, provider=cv2_tier.Cv2Tier())
    assert port_line["kind"] == "port" and cv_line["kind"] == "opencv"
    for line in (port_line, cv_line):
        assert line["os_cpu_count"] >= 1 and line["affinity_cpus"] >= 1 and "cpu_model" in line and line["cores"] == 2
        assert set(line["stage_s"]) == {"gray", "dis", "fit", "plan", "warp"}
    for k in ("transitions", "final", "frames", "masks", "counts", "gray", "flow"):
        assert np.array_equal(port[k], cv[k]), k


def test_primitive_parity_report_is_all_zero_under_the_standin(standin, oracle):
    from oracle import cv2_tier

    frames = np.stack([np.roll(synth_frames(1, 270, 480, seed=2)[0], (k, k), axis=(0, 1)) for k in range(3)])
    rep = cv2_tier.primitive_parity(oracle, cv2_tier.Cv2Tier(), frames, (240, 135), (0.5, 0.5, 0.5), pairs=2)
    assert rep["gray_u8_differing"] == 0 and rep["flow_epe_px"]["max"] == 0.0
    assert rep["warp_bilinear_q5"]["max"] == 0.0 and rep["warp_bicubic_q5"]["max"] == 0.0 and rep["mask_bilinear_differing"] == 0
    assert rep["warp_bilinear_exact"]["max"] > 0.0          # the report does tell the two sub-pixel conventions apart


def test_oracle_against_a_real_opencv(oracle):
    """UNREACHABLE TODAY (skips): the pinning test of DESIGN section 5.  Bounds = north_star's (pixels within 1e-3, flow
    within 1e-3 px) with the two named risks reported rather than hidden: an IPP cvtColor may move a few gray levels by
    one, and OpenCV >= 4.11 interpolates bilinear warps at unquantised coordinates (`exact`)."""
    from oracle import cv2_tier

    if not _real_cv2():
        pytest.skip("no real cv2 importable (parity with OpenCV stays unpinned: DESIGN section 5)")
    tier = cv2_tier.Cv2Tier()
    frames = np.stack([np.roll(synth_frames(1, 540, 960, seed=1)[0], (k, 2 * k), axis=(0, 1)) for k in range(5)])
    rep = cv2_tier.primitive_parity(oracle, tier, frames, None, (127 / 255.0,) * 3, pairs=4)
    print("cv2", cv2_tier.probe(), rep)
    assert rep["gray_u8_max_abs"] <= 1 and rep["gray_u8_differing"] <= 1e-3 * frames[..., 0].size
    assert rep["flow_epe_px_at_stride8_samples"]["max"] <= 1e-3
    assert min(rep["warp_bilinear_q5"]["max"], rep["warp_bilinear_exact"]["max"]) <= 1e-3
    assert rep["warp_bicubic_q5"]["max"] <= 1e-3 and rep["mask_bilinear_differing"] == 0
    assert np.nanmax(rep["fit_similarity_matrix_max_abs"]) <= 1e-4
