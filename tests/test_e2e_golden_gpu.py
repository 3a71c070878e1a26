"""The package's nodes against fixtures produced by the REFERENCE's own pipelines (tests/golden/make_e2e_golden.py:
`_stabilize_frames` of nodes/video_stabilizer_flow.py:213-640 and `apply_motion` of nodes/motion_apply.py:297-429 run
unmodified in the build container, their cv2 calls answered by the CPU oracle; likewise the Classic node's
`_stabilize_frames`, nodes/video_stabilizer_classic.py:163-568, and the Flow node on its phase-correlation fallback,
flow.py:90-130 -- the fixture's "estimator" field says which).

What this pins: the whole host chain of this build -- sticky mode walk (flow.py:324-346), working-size rescale, path /
target / diffs (flow.py:356-371), the crop_and_pad recentre incl. safe_region_* / center_offset (flow.py:500-529),
expand (flow.py:530-533), the keep_fov crop solver glue (flow.py:386-499, stabilizer_utils.py:448-837), mask statistics
and meta assembly (flow.py:596-640), the ProgressBar call sequence, Motion Apply's resolve / crop / expand / blur /
ticks -- and, through it, the HIP kernels' pixels, against the reference's control flow.  What it does not pin: the
OpenCV primitives themselves; on both sides those are the oracle's algorithm (parity against a real OpenCV stays
unpinned).

Tolerances (stated per check below):
  * structure, strings, ints, bools, None, list lengths, sizes, modes, progress calls: exact;
  * floats of the meta: |a-b| <= 1e-9 + 2e-6*|b|  (f32 fit matrices 2e-5 for perspective, where the reference side
    ran the oracle's LM refinement in another fp64 summation order than the kernel);
  * pixels: bit-equal whenever the applied matrices are bit-equal (always the case in translation mode: the median is
    exact); otherwise >= 99.9 % of the values within 1e-4 and none above 0.05 (a 1-ulp matrix difference can move a
    source coordinate across a 1/32-px quantisation step for a handful of pixels).
"""

import json
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
E2E = Path(__file__).parent / "golden" / "e2e"
INDEX = json.loads((E2E / "index.json").read_text())


def clip_frames(name, count=None):
    with np.load(E2E / "clips.npz") as z:
        u8 = z[name]
    if name == "big_half":
        u8 = np.repeat(np.repeat(u8, 2, axis=1), 2, axis=2)
    if count:
        u8 = u8[:count]
    return (u8.astype(np.float32) / np.float32(255.0)).astype(np.float32)


def assert_meta_close(got, want, path="meta", rel=2e-6, mat_tol=None):
    """Exact structure; floats within 1e-9 + rel*|want| (the f32 'matrix'/'applied_matrix' leaves within mat_tol)."""
    if isinstance(want, dict):
        assert isinstance(got, dict) and list(got.keys()) == list(want.keys()), f"{path}: keys {list(got.keys())} != {list(want.keys())}"
        for k in want:
            assert_meta_close(got[k], want[k], f"{path}.{k}", rel, mat_tol)
    elif isinstance(want, list):
        assert isinstance(got, list) and len(got) == len(want), f"{path}: length {len(got) if isinstance(got, list) else got} != {len(want)}"
        for i, (g, w) in enumerate(zip(got, want)):
            assert_meta_close(g, w, f"{path}[{i}]", rel, mat_tol)
    elif isinstance(want, bool) or want is None or isinstance(want, str):
        assert got == want and type(got) is type(want), f"{path}: {got!r} != {want!r}"
    elif isinstance(want, int):
        assert type(got) is int and got == want, f"{path}: {got!r} != {want!r}"
    else:
        assert isinstance(got, float), f"{path}: {type(got)} is not float"
        tol = 1e-9 + rel * abs(want)
        if mat_tol is not None and ("matrix" in path):
            tol = max(tol, mat_tol * max(1.0, abs(want)))
        assert abs(got - want) <= tol, f"{path}: {got!r} != {want!r} (tol {tol:g})"


def assert_pixels(got, want, exact, what):
    assert got.shape == want.shape and got.dtype == np.float32, f"{what}: shape {got.shape} vs {want.shape}"
    if exact:
        assert np.array_equal(got, want), f"{what}: max diff {np.abs(got - want).max()}"
        return 0.0
    d = np.abs(got - want)
    frac = float((d > 1e-4).mean())
    assert frac <= 1e-3 and float(d.max()) <= 0.05, f"{what}: {frac:.2e} of values off by > 1e-4, max {d.max():.3g}"
    return float(d.max())


@pytest.fixture(scope="module")
def api(pkg):
    from vstab_amd import apply_pipeline, comfy_compat, flow_pipeline, host_math

    class A:
        pass

    a = A()
    a.ap, a.fp, a.hm, a.cc = apply_pipeline, flow_pipeline, host_math, comfy_compat
    return a


@pytest.mark.parametrize("name", INDEX["flow"])
def test_flow_node_matches_reference_pipeline(api, ctx, monkeypatch, name):
    spec = json.loads((E2E / f"flow_{name}.json").read_text())
    want_meta = spec["meta"]
    frames = clip_frames(spec["clip"])
    bars = []

    class Bar(api.cc.ProgressBar):
        def __init__(self, total):
            super().__init__(total)
            self.calls = [("init", int(total))]
            bars.append(self)

        def update_absolute(self, value, total=None):
            self.calls.append((int(value), None if total is None else int(total)))

    monkeypatch.setattr(api.fp, "ProgressBar", Bar)
    a = spec["args"]
    # "classic": the Classic node's estimator (classic.py:69-160); "flow_phase_correlate": the Flow node with DIS
    # unavailable (flow.py:90-130), which this build selects through VSTAB_FLOW_BACKEND
    estimator = spec.get("estimator", "flow")
    if estimator == "flow_phase_correlate":
        monkeypatch.setenv("VSTAB_FLOW_BACKEND", "phase_correlate")
    res = api.fp._stabilize_frames(api.hm._normalize_video_input(frames), a[0], a[1], a[2], a[3], a[4], a[5], tuple(a[6]), a[7],
                                   estimator="classic" if estimator == "classic" else "flow")
    got_meta = json.loads(json.dumps(res.meta))
    perspective = a[1] == "perspective"
    assert_meta_close(got_meta, want_meta, mat_tol=2e-5 if perspective else 2e-6)
    # progress: the reference's update_absolute sequence (flow.py:282-287, 347-351, 589-593)
    assert len(bars) == 1 and [list(c) for c in bars[0].calls] == [list(c) for c in spec["progress"]]
    stride = spec["stride"]
    with np.load(E2E / f"flow_{name}.npz") as z:
        want_frames, want_masks = z["frames"], z["masks"]
    got_frames = np.asarray(res.frames, np.float32)
    got_masks = np.asarray(res.masks, np.float32)
    assert list(got_frames.shape) == spec["out_shape"] and got_masks.shape == got_frames.shape[:3] + (1,)
    same = (np.array([e["applied_matrix"] for e in got_meta["stabilization_warp"]["per_frame"]]).tobytes()
            == np.array([e["applied_matrix"] for e in want_meta["stabilization_warp"]["per_frame"]]).tobytes())
    if (a[1] == "translation" and estimator == "flow") or "bypass" in name:
        assert same, "translation mode: the applied matrices must be bit-equal to the reference's"
    assert_pixels(got_frames[:, ::stride, ::stride], want_frames, same, f"{name}.frames")
    assert_pixels(got_masks[:, ::stride, ::stride], want_masks, same, f"{name}.masks")


@pytest.mark.parametrize("name", INDEX["apply"])
def test_motion_apply_matches_reference_pipeline(api, ctx, name):
    spec = json.loads((E2E / f"apply_{name}.json").read_text())
    frames = clip_frames(spec["clip"], spec.get("clip_frames"))
    ticks = []
    res = api.ap.apply_motion(api.hm._normalize_video_input(frames), spec["meta_in"], tuple(spec["padding_rgb"]),
                              progress_callback=lambda: ticks.append(1), **spec["kwargs"])
    assert_meta_close(json.loads(json.dumps(res.meta)), spec["meta"])
    if spec["ticks"] is not None:
        assert len(ticks) == spec["ticks"]
    stride = spec["stride"]
    with np.load(E2E / f"apply_{name}.npz") as z:
        want_frames, want_masks = z["frames"], z["masks"]
    assert list(res.frames.shape) == spec["out_shape"]
    # the matrices come from the JSON on both sides -> identical inputs to the warp -> bit-equal pixels, also for the
    # 17- and 33-sample blur accumulation (same sample order, true division by float(S): motion_apply.py:171-200)
    assert_pixels(np.asarray(res.frames, np.float32)[:, ::stride, ::stride], want_frames, True, f"{name}.frames")
    assert_pixels(np.asarray(res.masks, np.float32)[:, ::stride, ::stride], want_masks, True, f"{name}.masks")
