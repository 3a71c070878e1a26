"""CPU side of the end-to-end goldens (tests/golden/e2e, written by make_e2e_golden.py from the reference's own
pipelines over oracle primitives):

  * the fixtures satisfy the reference's scripted properties (a sanity check of the cv2 stand-in they were made with):
    KA7 replay == stabilizer output bit for bit (check_crop_aspect_ratio.py:123-161), KA8 crop -> zero mask
    (check_crop_aspect_ratio.py:82-120), KA5 fallback, the sticky-mode walk of flow.py:338-339;
  * this build's HOST logic alone -- sticky selection, rescale, params, trajectory blend, framing, crop solver, meta --
    fed with the oracle's per-pair candidate fits on the same clips, reproduces the reference's matrices and meta.
    No GPU: the two device calls the planner makes (trajectory, crop coverage analysis) are answered by test doubles
    (NumPy / oracle).  The GPU suite (test_e2e_golden_gpu.py) repeats this through the HIP kernels, pixels included.
"""

import json
from pathlib import Path

import numpy as np
import pytest

from tests.test_distributed_cpu import NumpyTrajectoryCtx
from tests.test_e2e_golden_gpu import E2E, INDEX, assert_meta_close, clip_frames

pytestmark = []   # the helpers come from the GPU module; nothing here needs a GPU


def load(kind, name):
    spec = json.loads((E2E / f"{kind}_{name}.json").read_text())
    with np.load(E2E / f"{kind}_{name}.npz") as z:
        return spec, z["frames"], z["masks"]


def test_fixture_properties_of_the_reference_scripts():
    # sticky mode (flow.py:338-339): perspective -> similarity -> translation; similarity -> translation
    meta = load("flow", "broken_sticky_perspective")[0]["meta"]
    modes = [t["mode"] for t in meta["estimated_motion"]["per_transition"]]
    assert modes == ["perspective", "similarity", "similarity", "translation", "translation", "translation"]
    assert meta["transform_mode_applied"] == "translation" and meta["transform_mode_requested"] == "perspective"
    modes = [t["mode"] for t in load("flow", "broken_sticky_similarity")[0]["meta"]["estimated_motion"]["per_transition"]]
    assert modes == ["similarity"] * 3 + ["translation"] * 3
    # KA7: Motion Apply replay of the stabilizer's own meta is bit-identical
    _, f_stab, m_stab = load("flow", "tiny_expand_translation")
    _, f_rep, m_rep = load("apply", "tiny_replay_expand")
    assert np.array_equal(f_stab, f_rep) and np.array_equal(m_stab, m_rep)
    # KA8: crop framing leaves no padding and reports an aspect-preserving crop
    for name in ("mid_crop_keep_fov_0.6_similarity", "mid_crop_keep_fov_0_translation"):
        spec, _, masks = load("flow", name)
        fr = spec["meta"]["framing"]
        assert masks.max() == 0.0 and spec["meta"]["padding_fraction_max"] == 0.0 and fr["padding_detected"] is False
        assert abs(fr["crop_size"][0] / fr["crop_size"][1] - 96 / 64) < 1e-6
    for name in ("shake_crop_plain", "shake_crop_blur5_bicubic"):
        spec, _, masks = load("apply", name)
        assert masks.max() == 0.0 and "framing_fallback" not in spec["meta"] and spec["meta"]["motion_apply"]["framing_mode"] == "crop"
    # KA5
    spec, _, _ = load("apply", "tiny_crop_fallback")
    assert spec["meta"]["framing_fallback"] == "crop_and_pad" and spec["meta"]["motion_apply"]["framing_mode"] == "crop_and_pad"
    # clamps of motion_apply.py:315-318
    ma = load("apply", "shake_pad_alias_blur_clamped")[0]["meta"]["motion_apply"]
    assert ma["framing_mode"] == "crop_and_pad" and ma["motion_blur"] == 1.0 and ma["motion_blur_samples"] == 33
    # bypass keeps the frames
    spec, frames, masks = load("flow", "tiny_crop_keep_fov_bypass")
    assert np.array_equal(frames, clip_frames("tiny")) and not masks.any() and spec["meta"]["transform_mode_applied"] == "identity"


class OraclePlanCtx(NumpyTrajectoryCtx):
    """Planner test double: trajectory in NumPy, crop coverage analysis from the oracle."""

    def __init__(self, oracle):
        self.oracle = oracle

    def crop_analysis(self, matrices, src_size, out_size):
        return self.oracle.crop_analysis(matrices, src_size, out_size)


@pytest.mark.parametrize("name", [n for n in INDEX["flow"] if "bypass" not in n])
def test_host_planning_reproduces_reference_meta(pkg, oracle, name):
    from vstab_amd import flow_pipeline as fp
    from vstab_amd import host_math as hm

    spec = json.loads((E2E / f"flow_{name}.json").read_text())
    want = spec["meta"]
    framing, mode, lock, strength, smooth, keep_fov, rgb, fps = spec["args"]
    frames = clip_frames(spec["clip"])
    n, h, w, _ = frames.shape
    gray = oracle.gray_for_estimation(frames, hm._working_estimation_size(w, h))
    estimator = spec.get("estimator", "flow")
    if estimator == "classic":      # classic.py:69-160: corners of frame i tracked into frame i+1, fits on the tracked pairs
        records = []
        for i in range(n - 1):
            feats = oracle.good_features(gray[i], **oracle.GFTT)
            if feats.shape[0] < 12:
                records.append({})
                continue
            nxt, status = oracle.lk_track(gray[i], gray[i + 1], feats, **oracle.LK)
            records.append(oracle.fit_all_modes_points(feats, nxt, status, mode)[0])
    elif estimator == "flow_phase_correlate":   # flow.py:110-130: a translation per pair, confidence = peak response
        records = []
        for tx, ty, resp in oracle.phase_correlate_clip(gray):
            m = np.array([[1.0, 0.0, tx], [0.0, 1.0, ty], [0.0, 0.0, 1.0]], np.float32)
            records.append({"translation": {"matrix": m, "confidence": float(resp), "residual": 0.0, "accepted": True}})
    else:
        flow = oracle.dis_flow_clip(gray)
        records = [oracle.fit_all_modes(flow[i], 8, mode)[0] for i in range(n - 1)]
    plan = fp.plan_stabilization(OraclePlanCtx(oracle), records, (w, h), n, framing, mode, lock, strength, smooth, keep_fov,
                                 tuple(rgb), float(max(1.0, fps)), float(fps), estimator=estimator)
    meta = fp.prepare_meta(plan)
    # the padding statistics need the warp; everything else of flow.py:596-640 is host work
    for key in ("padding_fraction_mean", "padding_fraction_max"):
        meta[key] = want[key]
    meta["framing"]["padding_detected"] = want["framing"]["padding_detected"]
    # same key order as the reference once padding_detected is appended last (flow.py:596)
    assert_meta_close(json.loads(json.dumps(meta)), want, mat_tol=2e-5 if mode == "perspective" else 2e-6)
    # F11 (crop_and_pad recentre): final = T . M with T from the clip-global intersection of the warped frame boxes
    if framing == "crop_and_pad":
        fr = meta["framing"]
        apply_m = hm.params_to_matrices(np.array(want["estimated_motion"]["target_path"]) - np.array(want["estimated_motion"]["path"]), mode)
        shift = np.array([[1, 0, fr["center_offset"][0]], [0, 1, fr["center_offset"][1]], [0, 0, 1]], np.float32)
        recomposed = np.matmul(shift, apply_m)
        golden = np.array([e["applied_matrix"] for e in want["stabilization_warp"]["per_frame"]], np.float32)
        assert np.allclose(recomposed, golden, rtol=0, atol=2e-5)
        assert fr["safe_region_size"][0] <= w + 1e-9 and fr["safe_region_size"][1] <= h + 1e-9


@pytest.mark.skipif(not Path("/root/reference/scripts/check_motion_meta.py").exists(),
                    reason="the reference tree exists in the build container only")
def test_reference_check_scripts_pass_under_the_standin():
    """Build container only: the reference's OWN functional check scripts (scripts/check_motion_meta.py,
    check_crop_aspect_ratio.py, check_inverse_stabilization.py, check_node_schema.py) run against the reference's code with
    the oracle-backed cv2 stand-in that produced the e2e fixtures, in a child process.  They assert, among others:
    identity apply == input, blur determinism, tick counts N*S (+N), crop fallback, legacy inversion, crop aspect ratio
    and zero padding for Classic and Flow, Motion Apply replay of the stabilizer's meta bit-identical, the expand ->
    inverse round trip p99 <= 0.3 / mean <= 0.035.  If the stand-in misbehaved as a cv2, these would fail."""
    import subprocess
    import sys

    out = subprocess.run([sys.executable, str(Path(__file__).parent / "golden" / "run_reference_checks.py")], capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    for script in ("check_motion_meta.py", "check_node_schema.py", "check_crop_aspect_ratio.py", "check_inverse_stabilization.py"):
        assert f"{script}: exit code 0" in out.stdout, out.stdout[-2000:]
