"""End-to-end GPU tests of the two nodes through the C ABI: reference-pinned properties (KA1-KA12 of
SURVEY 8c) and parity of every stage against the oracle on a clip with known camera motion."""

import json
from pathlib import Path

import numpy as np
import pytest

from tests.util import synth_frames
from tests.util import test_matrices as make_matrices

pytestmark = pytest.mark.gpu
BORDER = np.array([127, 127, 127], np.float32) / 255.0


def ramp_frames(count, width=32, height=24):
    """Analytic ramp frames in the spirit of check_motion_meta.py:31-40 (own construction)."""
    yy, xx = np.mgrid[0:height, 0:width]
    out = np.zeros((count, height, width, 3), np.float32)
    for i in range(count):
        out[i, ..., 0] = (xx + i) / max(width + count - 1, 1)
        out[i, ..., 1] = yy / max(height - 1, 1)
        out[i, ..., 2] = ((xx + yy + i) % 7) / 6.0
    return out


@pytest.fixture(scope="module")
def api(pkg):
    from vstab_amd import apply_pipeline, flow_pipeline, host_math, meta_v2, nodes

    class A:
        pass

    a = A()
    a.ap, a.fp, a.hm, a.mv, a.nodes = apply_pipeline, flow_pipeline, host_math, meta_v2, nodes
    return a


def block(api, mats, size, out_size=None, n=None):
    return {"motion_meta": api.mv.build_motion_meta_v2(source="manual", frame_count=len(mats), fps=16.0, input_size=size,
                                                        output_size=out_size or size, matrices=mats)}


def shift(tx, ty):
    return np.array([[1, 0, tx], [0, 1, ty], [0, 0, 1]], np.float64)


def test_identity_apply_and_blur_zero_path(api, ctx):
    """KA1: identity == input within 1e-6, mask all 0, blur=0 bit-identical to the plain path."""
    frames = ramp_frames(3)
    c = api.hm._normalize_video_input(frames)
    meta = block(api, [np.eye(3)] * 3, (32, 24))
    base = api.ap.apply_motion(c, meta, (127, 127, 127))
    assert np.max(np.abs(base.frames - frames)) <= 1e-6 and base.masks.max() == 0.0 and base.masks.shape == (3, 24, 32, 1)
    again = api.ap.apply_motion(api.hm._normalize_video_input(frames), meta, (127, 127, 127), motion_blur=0.0, motion_blur_samples=17)
    assert np.array_equal(again.frames, base.frames) and np.array_equal(again.masks, base.masks)


def test_expand_enlarges_canvas(api, ctx):
    """KA2: I, T(6,-4), T(-6,4) on 32x24 -> 44x32 and meta says expand."""
    frames = ramp_frames(3)
    meta = block(api, [np.eye(3), shift(6, -4), shift(-6, 4)], (32, 24))
    r = api.ap.apply_motion(api.hm._normalize_video_input(frames), meta, (127, 127, 127), framing_mode="expand")
    assert r.frames.shape == (3, 32, 44, 3)
    assert r.meta["motion_apply"]["framing_mode"] == "expand" and r.meta["motion_apply"]["output_size"] == [44, 32]


def test_blur_deterministic_and_ticks(api, ctx):
    """KA3 + KA4: blur is deterministic, meta records it, progress ticks = N*S (+N for crop)."""
    frames = ramp_frames(4)
    meta = block(api, [shift(0.5 * i, -0.25 * i) for i in range(4)], (32, 24))
    ticks = []
    r1 = api.ap.apply_motion(api.hm._normalize_video_input(frames), meta, (127, 127, 127), motion_blur=0.5, motion_blur_samples=5,
                             progress_callback=lambda: ticks.append(1))
    r2 = api.ap.apply_motion(api.hm._normalize_video_input(frames), meta, (127, 127, 127), motion_blur=0.5, motion_blur_samples=5)
    assert np.array_equal(r1.frames, r2.frames) and r1.meta["motion_apply"]["motion_blur"] == 0.5
    assert r1.meta["motion_apply"]["motion_blur_samples"] == 5 and len(ticks) == 4 * 5
    ticks.clear()
    api.ap.apply_motion(api.hm._normalize_video_input(frames), meta, (127, 127, 127), framing_mode="crop", motion_blur=0.5,
                        motion_blur_samples=5, progress_callback=lambda: ticks.append(1))
    assert len(ticks) == 4 + 4 * 5
    # through the node: Ultra -> 33 samples
    import torch

    out = api.nodes.VideoStabilizerMotionApply.execute(torch.from_numpy(frames), meta, "crop_and_pad", "bilinear", "#7F7F7F", 0.5, "Ultra")
    assert out[2]["motion_apply"]["motion_blur_samples"] == 33 and out[2]["motion_apply"]["motion_blur_quality"] == "Ultra"
    out = api.nodes.VideoStabilizerMotionApply.execute(torch.from_numpy(frames), meta, "crop_and_pad", "bilinear", "#7F7F7F", 0.5, "bogus")
    assert out[2]["motion_apply"]["motion_blur_quality"] == "Standard"


def test_crop_fallback_and_crop_success(api, ctx):
    """KA5: a 60 px shift on 32x24 leaves no common region -> framing_fallback == crop_and_pad."""
    frames = ramp_frames(2)
    meta = block(api, [np.eye(3), shift(60, 0)], (32, 24))
    r = api.ap.apply_motion(api.hm._normalize_video_input(frames), meta, (127, 127, 127), framing_mode="crop")
    assert r.meta["framing_fallback"] == "crop_and_pad" and r.meta["motion_apply"]["framing_mode"] == "crop_and_pad"
    meta = block(api, [np.eye(3), shift(2, 1)], (32, 24))
    r = api.ap.apply_motion(api.hm._normalize_video_input(frames), meta, (127, 127, 127), framing_mode="crop")
    assert "framing_fallback" not in r.meta and r.masks.max() == 0.0 and r.meta["motion_apply"]["framing_mode"] == "crop"


def test_legacy_warp_resolution_and_context_selection(api, ctx):
    """KA6: stabilization_warp 80x50 -> 96x60: direct apply gives 60x96, inverse (context-selected) 50x80."""
    m = np.array([[1.2, 0, 0], [0, 1.2, 0], [0, 0, 1]], np.float32)
    warp = api.hm._build_stabilization_warp_meta(source_size=(80, 50), output_size=(96, 60), framing_mode="expand", applied_matrices=[m, m])
    meta = {"stabilization_warp": warp, "motion_meta": api.mv.applied_motion_meta_from_stabilization_warp(warp, 16.0, "estimated_flow")}
    direct = api.ap.apply_motion(api.hm._normalize_video_input(synth_frames(2, 50, 80)), meta, (0, 0, 0))
    assert direct.frames.shape == (2, 60, 96, 3) and direct.meta["motion_apply"]["source"] == "estimated_flow"
    inverse = api.ap.apply_motion(api.hm._normalize_video_input(synth_frames(2, 60, 96)), meta, (0, 0, 0))
    assert inverse.frames.shape == (2, 50, 80, 3) and inverse.meta["motion_apply"]["source"] == "legacy_stabilization"
    r = api.mv.resolve_motion_meta({"stabilization_warp": warp})
    assert r.input_size == (96, 60) and r.output_size == (80, 50) and np.allclose(r.per_frame[0].matrix, np.linalg.inv(m.astype(np.float64)))


def test_error_texts(api, ctx):
    """KA12 + A2: ValueError texts for missing meta, size and frame-count mismatches, bad enums."""
    frames = ramp_frames(2)
    c = api.hm._normalize_video_input(frames)
    with pytest.raises(ValueError, match="meta must contain motion_meta or stabilization_warp."):
        api.ap.apply_motion(c, {}, (0, 0, 0))
    meta = block(api, [np.eye(3)] * 2, (40, 24))
    with pytest.raises(ValueError, match=r"Input frames must match motion_meta.input_size \(40, 24\), got \(32, 24\)\."):
        api.ap.apply_motion(c, meta, (0, 0, 0))
    meta = block(api, [np.eye(3)] * 3, (32, 24))
    with pytest.raises(ValueError, match=r"Frame count mismatch: got 2 frame\(s\), metadata has 3 matrix entry/entries\."):
        api.ap.apply_motion(c, meta, (0, 0, 0))
    meta = block(api, [np.eye(3)] * 2, (32, 24))
    with pytest.raises(ValueError, match="Unsupported interpolation 'nearest'; expected 'bilinear' or 'bicubic'."):
        api.ap.apply_motion(c, meta, (0, 0, 0), interpolation="nearest")
    with pytest.raises(ValueError, match="Unsupported framing_mode 'zoom'; expected 'crop_and_pad', 'crop', or 'expand'."):
        api.ap.apply_motion(c, meta, (0, 0, 0), framing_mode="zoom")
    r = api.ap.apply_motion(c, meta, (0, 0, 0), framing_mode="pad")
    assert r.meta["motion_apply"]["framing_mode"] == "crop_and_pad"


ANALYTIC_CORNER_PX, ANALYTIC_LIN = 0.3, 3e-4   # see the comment at their use (measured max: 0.21 px / 2.0e-4, the 1080p case)

FLOW_META_KEYS = ["frames", "transform_mode_requested", "transform_mode_applied", "camera_lock", "strength", "strength_effective",
                  "smooth", "fps_requested", "fps_effective", "framing", "keep_fov_applied", "padding_color_rgb", "flow_backend",
                  "flow_fallback_reason", "stabilization_warp", "estimated_motion", "padding_fraction_mean", "padding_fraction_max",
                  "motion_meta"]


@pytest.mark.parametrize("size,mode,framing", [((480, 270), "similarity", "crop_and_pad"), ((480, 270), "translation", "expand"),
                                               ((1920, 1080), "similarity", "crop_and_pad"),
                                               ((480, 270), "perspective", "crop_and_pad"),
                                               ((1000, 777), "similarity", "expand"),        # non-integer area ratio 960x746
                                               ((720, 1280), "similarity", "crop_and_pad")])  # portrait, working size 540x960
def test_flow_pipeline_against_oracle(api, ctx, oracle, size, mode, framing):
    """Every stage of the Flow node vs the oracle on a clip with known motion; KA7 replay bit-identity;
    the meta key set / list lengths / types of SURVEY 8a "Meta detail"."""
    from tests.test_dis_gpu import moving_clip

    w, h = size
    n = 7 if w <= 960 else 4
    gray_full, params = moving_clip(n, h, w, seed=w)
    frames = np.repeat(gray_full[..., None].astype(np.float32) / 255.0, 3, axis=-1)
    frames[..., 1] *= 0.9
    frames = np.ascontiguousarray(frames)
    c = api.hm._normalize_video_input(frames)
    res = api.fp._stabilize_frames(c, framing, mode, False, 0.7, 0.5, 0.6, (127, 127, 127), 16.0)
    meta = res.meta
    assert list(meta.keys()) == FLOW_META_KEYS
    assert meta["frames"] == n and meta["flow_backend"] == "DIS" and meta["flow_fallback_reason"] is None
    em = meta["estimated_motion"]
    pdim = {"translation": 2, "similarity": 4, "perspective": 8}[mode]
    assert len(em["per_transition"]) == n - 1 and [t["index"] for t in em["per_transition"]] == list(range(n - 1))
    assert np.array(em["path"]).shape == (n, pdim) and np.array(em["target_path"]).shape == (n, pdim)
    assert len(meta["stabilization_warp"]["per_frame"]) == n and meta["motion_meta"]["frame_count"] == n
    assert meta["motion_meta"]["version"] == 2 and meta["motion_meta"]["source"] == "estimated_flow"
    json.dumps(meta)  # plain JSON types only
    # --- estimation stages vs oracle
    work = api.hm._working_estimation_size(w, h)
    g = oracle.gray_for_estimation(frames, work)
    flow = oracle.dis_flow_clip(g)
    recs = []
    for i in range(n - 1):
        r, nv, nt = oracle.fit_all_modes(flow[i], 8, mode)
        recs.append(r)
    mats, modes, confs, resids, active = api.fp.select_transitions(recs, mode)
    assert meta["transform_mode_applied"] == active
    for i, t in enumerate(em["per_transition"]):
        assert t["mode"] == modes[i] and t["confidence"] == confs[i]
        full = api.hm._rescale_transform_to_full(mats[i], (w, h), work) if work else mats[i]
        tol = dict(rtol=2e-5, atol=1e-6) if mode == "perspective" else dict(rtol=0, atol=2e-5 if work else 1e-6)
        assert np.allclose(np.array(t["matrix"], np.float32), full, **tol)
        assert t["residual"] == pytest.approx(resids[i], rel=1e-6)
    # known motion is recovered (independent of the oracle).  moving_clip() quantises an analytic texture to u8 gray and
    # this test runs at working sizes down to 480x270 with a 48-wave texture; the tight bounds (<= 0.05 px, <= 1e-4) live
    # in tests/test_analytic_gpu.py on float clips -- here: 0.3 px anywhere in the frame (worst case: the 1080p clip, whose
    # texture reaches a 12-px wavelength at working resolution after the 2x area downscale)
    def to_texture(pr):  # frame coords -> texture coords of moving_clip()
        tx, ty, th, sc = pr
        c, sn = np.cos(th) / sc, np.sin(th) / sc
        lin = np.array([[c, sn, 0], [-sn, c, 0], [0, 0, 1.0]])
        pre = np.array([[1, 0, -w / 2 - tx], [0, 1, -h / 2 - ty], [0, 0, 1.0]])
        post = np.array([[1, 0, w / 2], [0, 1, h / 2], [0, 0, 1.0]])
        return post @ lin @ pre

    if mode != "translation":   # a translation model cannot represent the clip's rotation / zoom
        import bench

        cam = np.stack([np.linalg.inv(to_texture(p)) for p in params])
        acc = bench.transition_accuracy([t["matrix"] for t in em["per_transition"]], cam, (w, h), work)
        assert acc["corner_px"]["max"] < ANALYTIC_CORNER_PX and acc["lin_2x2"]["max"] < ANALYTIC_LIN, acc
    # --- warp vs oracle with the node's own matrices
    fm = np.array([e["applied_matrix"] for e in meta["stabilization_warp"]["per_frame"]], np.float32)
    out_size = tuple(meta["stabilization_warp"]["output_size"])
    ref, ref_mask, cnt = oracle.warp_clip(frames, fm, out_size, border=BORDER)
    assert np.array_equal(res.frames, ref) and np.array_equal(res.masks[..., 0], ref_mask)
    ratios = [float(np.float32(k) / np.float32(out_size[0] * out_size[1])) for k in cnt]
    assert meta["padding_fraction_mean"] == float(np.mean(ratios)) and meta["padding_fraction_max"] == float(np.max(ratios))
    assert meta["framing"]["padding_detected"] == bool(cnt.max() > 0)
    if framing == "expand":
        assert meta["framing"]["expanded_size"] == list(out_size) and res.frames.shape[1:3] == (out_size[1], out_size[0])
    # --- KA7: Motion Apply replay of the stabilizer's own meta is bit-identical
    replay = api.ap.apply_motion(api.hm._normalize_video_input(frames), meta, (127, 127, 127), framing_mode="crop_and_pad")
    assert np.array_equal(replay.frames, res.frames) and np.array_equal(replay.masks, res.masks)


def test_flow_node_small_paths(api, ctx):
    """F15: single frame passthrough; camera_lock; crop bypass at keep_fov ~ 1; dict input keeps its template."""
    import torch

    frames = synth_frames(1, 60, 80)
    out = api.nodes.VideoStabilizerFlow.execute(torch.from_numpy(frames), 16.0, "crop_and_pad", "similarity", False, 0.7, 0.5, 0.6, "#7F7F7F")
    assert np.array_equal(out[0].numpy(), frames) and out[2]["note"].startswith("Single-frame") and out[2]["motion_meta"]["frame_count"] == 1
    frames = synth_frames(5, 136, 240)
    out = api.nodes.VideoStabilizerFlow.execute({"frames": torch.from_numpy(frames), "fps": 24.0, "tag": "x"}, 0.0, "crop", "similarity", True, 0.7,
                                                0.5, 1.0, "#102030")
    assert isinstance(out[0], dict) and out[0]["tag"] == "x" and np.array_equal(out[0]["frames"].numpy(), frames)
    assert out[2]["fps_effective"] == 24.0 and out[2]["fps_requested"] is None and out[2]["transform_mode_applied"] == "identity"
    out = api.nodes.VideoStabilizerFlow.execute(torch.from_numpy(frames), 16.0, "crop_and_pad", "similarity", True, 0.7, 0.2, 0.6, "#7F7F7F")
    assert out[2]["smooth"] == 0.85 and np.all(np.array(out[2]["estimated_motion"]["target_path"]) == 0.0)
    out = api.nodes.VideoStabilizerFlow.execute(torch.from_numpy(frames), 16.0, "crop", "similarity", False, 0.7, 0.5, 0.6, "#7F7F7F")
    assert out[2]["framing"]["mode"] == "crop" and float(out[1].max()) == 0.0


def test_expand_then_inverse_round_trip(api, ctx):
    """KA9 (check_inverse_stabilization.py:161-165): expand -> inverse restores the clip, p99 <= 0.3, mean <= 0.035;
    also the device-resident chaining switch (N3)."""
    import os

    import torch

    from tests.test_dis_gpu import moving_clip

    gray, _ = moving_clip(6, 270, 480, seed=21)
    frames = np.ascontiguousarray(np.repeat(gray[..., None].astype(np.float32) / 255.0, 3, axis=-1))
    out = api.nodes.VideoStabilizerFlow.execute(torch.from_numpy(frames), 16.0, "expand", "similarity", False, 0.7, 0.5, 0.6, "#7F7F7F")
    stab, meta = out[0], out[2]
    inv = api.nodes.VideoStabilizerInverse.execute(stab, meta, "#7F7F7F")
    restored, mask, imeta = inv[0].numpy(), inv[1].numpy(), inv[2]
    assert restored.shape == frames.shape and restored.dtype == np.float32 and mask.shape == frames.shape[:3]
    valid = mask < 0.5
    err = np.abs(restored - frames)[valid]
    assert np.percentile(err, 99) <= 0.3 and err.mean() <= 0.035
    assert imeta["inverse_stabilization"]["output_size"] == [480, 270] and "motion_apply" not in imeta
    assert imeta["motion_meta"] == meta["motion_meta"]
    os.environ["VSTAB_KEEP_ON_DEVICE"] = "1"
    try:
        out = api.nodes.VideoStabilizerFlow.execute(torch.from_numpy(frames), 16.0, "expand", "similarity", False, 0.7, 0.5, 0.6, "#7F7F7F")
        assert out[0].is_cuda and out[1].is_cuda and out[1].ndim == 3
        chained = api.nodes.VideoStabilizerMotionApply.execute(out[0], {"stabilization_warp": meta["stabilization_warp"]}, "crop_and_pad",
                                                               "bilinear", "#7F7F7F", 0.0, "Standard")
        assert chained[0].is_cuda and np.array_equal(chained[0].cpu().numpy(), restored)
    finally:
        del os.environ["VSTAB_KEEP_ON_DEVICE"]


@pytest.mark.parametrize("framing,mode,keep_fov", [("crop_and_pad", "similarity", 0.6), ("expand", "translation", 0.6),
                                                   ("crop", "translation", 1.0)])
def test_reference_script_scenarios_on_tiny_clip(api, ctx, oracle, framing, mode, keep_fov):
    """The three Flow scenarios of scripts/compare_refactor_behavior.py:380-393 on an 8-frame 73x45 clip at 24 fps
    (DIS auto-selects its scales at this size); every stage is checked against the oracle."""
    yy, xx = np.mgrid[0:45, 0:73].astype(np.float32)
    frames = np.zeros((8, 45, 73, 3), np.float32)
    for i in range(8):
        sx, sy = xx - 0.7 * i, yy - 0.35 * i
        frames[i, ..., 0] = 0.5 + 0.4 * np.sin(sx * 0.31) * np.cos(sy * 0.23)
        frames[i, ..., 1] = 0.5 + 0.4 * np.cos(sx * 0.17 + sy * 0.29)
        frames[i, ..., 2] = ((np.floor(sx / 5) + np.floor(sy / 7)) % 2) * 0.8 + 0.1
    res = api.fp._stabilize_frames(api.hm._normalize_video_input(frames), framing, mode, False, 0.7, 0.5, keep_fov, (127, 127, 127), 24.0)
    if framing == "crop":
        assert np.array_equal(res.frames, frames) and res.meta["note"].startswith("keep_fov~=1.0") and res.masks.max() == 0.0
        return
    g = oracle.gray_for_estimation(frames, None)
    flow = oracle.dis_flow_clip(g)
    recs = [oracle.fit_all_modes(flow[i], 8, mode)[0] for i in range(7)]
    mats, modes, confs, resids, active = api.fp.select_transitions(recs, mode)
    for i, t in enumerate(res.meta["estimated_motion"]["per_transition"]):
        assert t["mode"] == modes[i] and t["confidence"] == confs[i]
        assert np.allclose(np.array(t["matrix"], np.float32), mats[i], rtol=0, atol=1e-6)
    fm = np.array([e["applied_matrix"] for e in res.meta["stabilization_warp"]["per_frame"]], np.float32)
    out_size = tuple(res.meta["stabilization_warp"]["output_size"])
    ref, ref_mask, _ = oracle.warp_clip(frames, fm, out_size, border=BORDER)
    assert np.array_equal(res.frames, ref) and np.array_equal(res.masks[..., 0], ref_mask)
    assert res.meta["fps_effective"] == 24.0


def test_pinned_output_option_is_transparent(api, ctx, monkeypatch):
    """VSTAB_PINNED_OUTPUT=1 only changes where the returned CPU tensors live (page-locked memory, one direct DMA):
    same values, same shapes, still CPU tensors."""
    import torch

    frames = torch.from_numpy(synth_frames(4, 72, 128))
    args = (frames, 16.0, "crop_and_pad", "similarity", False, 0.7, 0.5, 0.6, "#7F7F7F")
    plain = api.nodes.VideoStabilizerFlow.execute(*args)
    monkeypatch.setenv("VSTAB_PINNED_OUTPUT", "1")
    pinned = api.nodes.VideoStabilizerFlow.execute(*args)
    assert pinned[0].device.type == "cpu" and pinned[0].is_pinned() and pinned[1].is_pinned() and not plain[0].is_pinned()
    assert torch.equal(plain[0], pinned[0]) and torch.equal(plain[1], pinned[1]) and plain[2] == pinned[2]


def test_value_range_sniff_is_settled_on_the_gpu(api, ctx):
    """F0 (stabilizer_utils.py:127-131): float frames whose maximum exceeds 1.5 are 0..255 data and get divided by 255,
    decided per frame.  The nodes take the per-frame maxima from their first pass over the pixels on the GPU (gray
    kernel / vstab_frame_range) and repeat the estimation on the rescaled clip when needed: a clip with two 0..255
    frames must give exactly the result of the same clip rescaled by the caller, and the caller's tensor stays untouched."""
    import torch

    from tests.test_dis_gpu import moving_clip

    gray, _ = moving_clip(6, 136, 240, seed=77)
    clean = np.ascontiguousarray(np.repeat(gray[..., None].astype(np.float32) / 255.0, 3, axis=-1))
    mixed = clean.copy()
    mixed[1] = mixed[1] * 255.0
    mixed[4] = mixed[4] * 255.0
    # the reference divides float32 by 255.0: the rescaled frames are (x*255)/255, not bit-equal to x
    expect_in = clean.copy()
    expect_in[1] = mixed[1] / np.float32(255.0)
    expect_in[4] = mixed[4] / np.float32(255.0)
    args = (16.0, "crop_and_pad", "similarity", False, 0.7, 0.5, 0.6, "#7F7F7F")
    for device in ("cpu", "cuda"):
        t = torch.from_numpy(mixed.copy()).to(device)
        keep = t.clone()
        got = api.nodes.VideoStabilizerFlow.execute(t, *args)
        want = api.nodes.VideoStabilizerFlow.execute(torch.from_numpy(expect_in).to(device), *args)
        assert torch.equal(t, keep), "inputs are read-only views (SURVEY 8b: ownership)"
        assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1]) and got[2] == want[2]
    blockmeta = got[2]
    t = torch.from_numpy(mixed.copy())
    a = api.nodes.VideoStabilizerMotionApply.execute(t, blockmeta, "expand", "bicubic", "#102030", 0.5, "Draft")
    b = api.nodes.VideoStabilizerMotionApply.execute(torch.from_numpy(expect_in), blockmeta, "expand", "bicubic", "#102030", 0.5, "Draft")
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and a[2] == b[2]
    # single-frame passthrough settles the sniff on the host
    one = api.nodes.VideoStabilizerFlow.execute(torch.from_numpy(mixed[1:2].copy()), *args)
    assert np.array_equal(one[0].numpy(), expect_in[1:2])


def test_upload_is_ordered_behind_running_kernels(ctx, api):
    """ADVICE r2 (vstab_xfer.hip): an upload's destination may be a block the caching allocator handed back while a
    kernel queued on the context's stream still reads it.  Clip A is blur-warped from host memory with the outputs kept
    on the device; its device frames are dropped as apply_motion returns (the 17-sample warp is still running), and clip
    B of the same size is uploaded at once -- it lands in A's block.  A's result must equal the result of a run that
    synchronises before anything is reused."""
    import torch

    n, h, w = 6, 1080, 1920
    a = torch.from_numpy(synth_frames(1, h, w, seed=3)).expand(n, h, w, 3).contiguous()
    b = torch.zeros_like(a)
    mats = make_matrices(n, w, h, "similarity", seed=4)
    meta = block(api, list(mats), (w, h))

    def run(sync_first):
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        r = api.ap.apply_motion(api.hm._normalize_video_input(a), meta, (127, 127, 127), framing_mode="crop_and_pad",
                                interpolation="bicubic", motion_blur=0.5, motion_blur_samples=17, ctx=ctx, keep_on_device=True)
        if sync_first:
            torch.cuda.synchronize()
        other = ctx.upload(b)            # same size as A's freed device frames: the allocator returns that block
        out = r.frames.cpu()
        del other
        return out

    want = run(True)
    got = run(False)
    assert torch.equal(got, want)


def test_transfers_survive_a_refused_helper_thread(pkg):
    """a box that refuses a std::thread must not take the process down (exception through extern "C",
    helpers parked in a barrier): the team shrinks to the threads that exist, down to the caller alone.  The refusal is
    injected by the TEST build of the library (VSTAB_DEBUG_XFER_SPAWN_FAIL; the shipped one has no such knob)."""
    from tests.util import run_with_hooks

    run_with_hooks("""
        ctx = native.Context(0)
        host = torch.arange((40 << 20) // 4, dtype=torch.int32)
        for k in ("0", "1", "5"):
            os.environ["VSTAB_DEBUG_XFER_SPAWN_FAIL"] = k
            dev = ctx.upload(host)
            assert torch.equal(ctx.download(dev), host), k
    """)


class _Interrupted(Exception):
    """Stand-in for comfy.model_management.InterruptProcessingException."""


class _ManagementStub:
    """comfy.model_management as the nodes see it: raises at the k-th poll (k = 1: the first one)."""

    def __init__(self, raise_at):
        self.raise_at, self.polls = raise_at, 0

    def throw_exception_if_processing_interrupted(self):
        self.polls += 1
        if self.polls == self.raise_at:
            raise _Interrupted()


@pytest.mark.parametrize("raise_at", [1, 2])
def test_cancel_leaves_both_nodes_reusable(api, ctx, monkeypatch, raise_at):
    """Cooperative cancel (flow.py:275-277, 352, 594): ComfyUI's interrupt surfaces as an exception from
    throw_exception_if_processing_interrupted().  It must propagate out of both nodes from every poll site, and the
    shared context must compute the same results afterwards (no stream left waiting, no half-consumed status)."""
    import torch
    from vstab_amd import comfy_compat

    frames = torch.from_numpy(synth_frames(6, 136, 240, seed=11))
    flow_args = (frames, 16.0, "crop_and_pad", "similarity", False, 0.7, 0.5, 0.6, "#7F7F7F")
    clean = api.nodes.VideoStabilizerFlow.execute(*flow_args)
    apply_args = (frames, {"motion_meta": clean[2]["motion_meta"]}, "crop", "bicubic", "#7F7F7F", 0.5, "High")
    clean_apply = api.nodes.VideoStabilizerMotionApply.execute(*apply_args)

    for node, args, want in ((api.nodes.VideoStabilizerFlow, flow_args, clean), (api.nodes.VideoStabilizerMotionApply, apply_args, clean_apply)):
        stub = _ManagementStub(raise_at)
        for mod in (comfy_compat, api.fp, api.ap):
            if hasattr(mod, "model_management"):
                monkeypatch.setattr(mod, "model_management", stub)
        with pytest.raises(_Interrupted):
            node.execute(*args)
        assert stub.polls == raise_at
        monkeypatch.setattr(comfy_compat, "model_management", None)
        again = node.execute(*args)
        assert torch.equal(again[0], want[0]) and torch.equal(again[1], want[1]) and again[2] == want[2]
    # the polls of a whole run: Flow polls after the estimation and after the warp; Motion Apply (crop) before the
    # coverage pass, before the warp launch and once after it
    counter = _ManagementStub(-1)
    monkeypatch.setattr(comfy_compat, "model_management", counter)
    api.nodes.VideoStabilizerFlow.execute(*flow_args)
    assert counter.polls >= 2
    counter.polls = 0
    api.nodes.VideoStabilizerMotionApply.execute(*apply_args)
    assert counter.polls >= 3
