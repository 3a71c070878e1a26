"""BASELINE configs C3 and C5 exercised AT THEIR FRAME SIZES as node chains, against the oracle.

  C3  1920x1080: Video Stabilizer Flow (perspective, crop_and_pad) -> Video Stabilizer Motion Apply on the ORIGINAL
      frames with the returned meta (crop_and_pad, bicubic, motion_blur 0.5, "High" = 17 samples)
  C5  3840x2160: Flow (similarity, expand) -> Motion Apply (expand, bilinear, motion_blur 0.5, "Ultra" = 33 samples)

Call shapes: nodes/video_stabilizer_flow.py:734-763, nodes/video_stabilizer_motion_apply.py:86-129.  The clips are a
few frames of the bench's synthetic clip so that the scalar oracle finishes in seconds; BASELINE's frame counts (256 /
512) only repeat the same per-frame work.  Checks: every estimation stage of the Flow node against the oracle on the
same frames (gray / DIS bit-exact through the fits: inlier ratios exact, matrices at the stated tolerance), the Flow
node's pixels against the oracle warp of its own matrices (bit-exact), and the Motion Apply output against the
oracle's S-sample blur of the same JSON matrices (bit-exact, frames and soft masks).
"""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
BORDER = np.array([127, 127, 127], np.float32) / 255.0


@pytest.fixture(scope="module")
def api(pkg):
    from vstab_amd import flow_pipeline, host_math, nodes

    class A:
        pass

    a = A()
    a.fp, a.hm, a.nodes = flow_pipeline, host_math, nodes
    return a


def bench_clip(n, h, w):
    import torch

    import bench

    return bench.synth_clip(n, 0, h, w, torch.device("cuda")).cpu()


def check_flow_stage(api, oracle, frames, meta, mode):
    n, h, w, _ = frames.shape
    work = api.hm._working_estimation_size(w, h)
    assert work == (960, 540)
    gray = oracle.gray_for_estimation(frames, work)
    flow = oracle.dis_flow_clip(gray)
    recs = [oracle.fit_all_modes(flow[i], 8, mode)[0] for i in range(n - 1)]
    mats, modes, confs, resids, active = api.fp.select_transitions(recs, mode)
    assert meta["transform_mode_applied"] == active == mode
    for i, t in enumerate(meta["estimated_motion"]["per_transition"]):
        assert t["mode"] == modes[i] and t["confidence"] == confs[i]          # inlier counts are integers: exact
        full = api.hm._rescale_transform_to_full(mats[i], (w, h), work)
        got = np.array(t["matrix"], np.float32)
        # f32 matrices; perspective: the kernel's LM refinement sums its normal equations in another fp64 order
        tol = dict(rtol=2e-5, atol=2e-5) if mode == "perspective" else dict(rtol=0, atol=2e-5)
        assert np.allclose(got, full, **tol), (i, np.abs(got - full).max())
        assert t["residual"] == pytest.approx(resids[i], rel=1e-5)


def test_c3_flow_perspective_then_motion_apply_bicubic_blur_high_1080p(api, ctx, oracle):
    frames_t = bench_clip(4, 1080, 1920)
    frames = frames_t.numpy()
    out = api.nodes.VideoStabilizerFlow.execute(frames_t, 16.0, "crop_and_pad", "perspective", False, 0.7, 0.5, 0.6, "#7F7F7F")
    stab, mask, meta = out[0].numpy(), out[1].numpy(), out[2]
    assert stab.shape == (4, 1080, 1920, 3) and meta["framing"]["mode"] == "crop_and_pad"
    check_flow_stage(api, oracle, frames, meta, "perspective")
    fm = np.array([e["applied_matrix"] for e in meta["stabilization_warp"]["per_frame"]], np.float32)
    assert np.abs(fm[:, 2, :2]).max() > 0.0, "the perspective rows are exercised"
    ref, ref_mask, cnt = oracle.warp_clip(frames, fm, (1920, 1080), border=BORDER)
    assert np.array_equal(stab, ref) and np.array_equal(mask, ref_mask)
    ratios = [float(np.float32(k) / np.float32(1920 * 1080)) for k in cnt]
    assert meta["padding_fraction_mean"] == float(np.mean(ratios)) and meta["padding_fraction_max"] == float(np.max(ratios))

    applied = api.nodes.VideoStabilizerMotionApply.execute(frames_t, meta, "crop_and_pad", "bicubic", "#7F7F7F", 0.5, "High")
    a_frames, a_mask, a_meta = applied[0].numpy(), applied[1].numpy(), applied[2]
    ma = a_meta["motion_apply"]
    assert ma == {"input_size": [1920, 1080], "output_size": [1920, 1080], "framing_mode": "crop_and_pad", "interpolation": "bicubic",
                  "motion_blur": 0.5, "motion_blur_samples": 17, "source": "estimated_flow", "motion_blur_quality": "High"}
    m64 = np.array([e["matrix"] for e in meta["motion_meta"]["per_frame"]], np.float64)
    assert np.array_equal(m64.astype(np.float32), fm)
    ref, ref_mask = oracle.warp_blur_clip(frames, m64, (1920, 1080), 0.5, 17, interp="bicubic", border=BORDER)
    assert np.array_equal(a_frames, ref), f"max diff {np.abs(a_frames - ref).max()}"
    assert np.array_equal(a_mask, ref_mask)
    assert a_mask.min() >= 0.0 and a_mask.max() <= 1.0 and len(np.unique(a_mask)) > 2, "soft mask: k/17 coverage levels"


def test_c5_flow_expand_then_motion_apply_expand_blur_ultra_4k(api, ctx, oracle):
    frames_t = bench_clip(3, 2160, 3840)
    frames = frames_t.numpy()
    out = api.nodes.VideoStabilizerFlow.execute(frames_t, 16.0, "expand", "similarity", False, 0.7, 0.5, 0.6, "#7F7F7F")
    stab, mask, meta = out[0].numpy(), out[1].numpy(), out[2]
    ow, oh = meta["framing"]["expanded_size"]
    assert (ow, oh) != (3840, 2160) and ow >= 3840 and oh >= 2160 and stab.shape == (3, oh, ow, 3)
    assert meta["stabilization_warp"]["output_size"] == [ow, oh] and meta["motion_meta"]["output_size"] == [ow, oh]
    check_flow_stage(api, oracle, frames, meta, "similarity")
    fm = np.array([e["applied_matrix"] for e in meta["stabilization_warp"]["per_frame"]], np.float32)
    ref, ref_mask, _ = oracle.warp_clip(frames, fm, (ow, oh), border=BORDER)
    assert np.array_equal(stab, ref) and np.array_equal(mask, ref_mask)
    del ref, ref_mask, stab, mask

    # Motion Apply on the original 4K frames with the returned meta, expand framing: the canvas is re-derived from the
    # motion matrices (motion_apply.py:288-294) -- the same bounding boxes, so the same canvas
    applied = api.nodes.VideoStabilizerMotionApply.execute(frames_t, meta, "expand", "bilinear", "#7F7F7F", 0.5, "Ultra")
    a_frames, a_mask, a_meta = applied[0].numpy(), applied[1].numpy(), applied[2]
    ma = a_meta["motion_apply"]
    assert ma["framing_mode"] == "expand" and ma["motion_blur_samples"] == 33 and ma["motion_blur_quality"] == "Ultra"
    ew, eh = ma["output_size"]
    assert a_frames.shape == (3, eh, ew, 3)
    m64 = [np.array(e["matrix"], np.float64) for e in meta["motion_meta"]["per_frame"]]
    mins, maxs = api.hm._compute_bounding_boxes(m64, 3840, 2160)
    shift, size = api.hm._prepare_expand_transform(mins, maxs)
    assert list(size) == [ew, eh]
    expanded = np.stack([shift @ m for m in m64])
    ref, ref_mask = oracle.warp_blur_clip(frames, expanded, (ew, eh), 0.5, 33, interp="bilinear", border=BORDER)
    assert np.array_equal(a_frames, ref), f"max diff {np.abs(a_frames - ref).max()}"
    assert np.array_equal(a_mask, ref_mask)


def test_c2_at_baseline_frame_count_matches_oracle(api, ctx):
    """BASELINE configs[1] AT ITS FRAME COUNT: the 256 x 1080p bench clip through
    `_stabilize_frames` exactly as bench.py's timed step runs it, against the CPU oracle run over the same 256 frames:
    all 255 reported transitions, confidences, final matrices, every output pixel, every mask pixel and the padding
    statistics bit-equal; then HIP against HIP: pairs {0,127,254} as 2-frame clips (the split DIS launch form) and
    frames {0,127,255} warped alone equal the whole-clip run (fused form chosen because 8 P >= 7 CUs, XCD block remap over
    130 k blocks, 64-bit frame bases, the 628 MB workspace carve).  Bounds: none -- equality."""
    import os

    import torch

    import bench

    n, h, w = 256, 1080, 1920
    frames = bench.synth_clip(n, 0, h, w, torch.device("cuda"))
    res = api.fp._stabilize_frames(api.hm._normalize_video_input(frames), *bench.FLOW_ARGS, ctx=ctx, keep_on_device=True)
    meta = res.meta
    assert meta["frames"] == n and meta["transform_mode_applied"] == "similarity"
    inv = bench.check_batch_invariance(api.fp, api.hm, ctx, frames, meta, res.frames)
    assert inv["fit_records_equal"] and inv["frames_equal"], inv
    host = frames.cpu().numpy()
    _, port = bench.cpu_baseline(host, min(16, len(os.sched_getaffinity(0))), keep_outputs=True)
    port["source"] = host
    chk = bench.check_against_oracle(meta, res.frames, res.masks, port)
    assert chk["bit_equal"], chk
    acc = bench.transition_accuracy([t["matrix"] for t in meta["estimated_motion"]["per_transition"]],
                                    bench.camera_matrices(n, 0, w, h), (w, h), (960, 540))
    # oracle-independent: the analytic motion of the bench clip is recovered (bounds: tests/test_analytic_gpu.py)
    assert acc["corner_px"]["max"] < 0.1 and acc["lin_2x2"]["max"] < 2e-4, acc


def test_c3_at_baseline_frame_count_sampled_frames_match_oracle(api, ctx, oracle):
    """BASELINE configs[2] at 256 x 1080p: Flow perspective -> Motion Apply bicubic, blur 0.5, High (17 samples),
    device-resident.  The Flow stage is checked on all 255 pairs against the oracle; the blurred output is checked
    bit-exactly on frames {0, 1, 127, 254, 255}: a blurred frame depends on its own pixels and on its own and its
    successor's matrix (the last frame: its predecessor's; motion_apply.py:125-134), so the oracle renders each sampled
    frame from a 2-frame window of the clip."""
    import torch

    import bench
    from vstab_amd import apply_pipeline as ap

    n, h, w = 256, 1080, 1920
    frames = bench.synth_clip(n, 0, h, w, torch.device("cuda"))
    res = api.fp._stabilize_frames(api.hm._normalize_video_input(frames), "crop_and_pad", "perspective", False, 0.7, 0.5, 0.6,
                                   (127, 127, 127), 16.0, ctx=ctx, keep_on_device=True)
    meta = res.meta
    host = frames.cpu().numpy()
    check_flow_stage(api, oracle, host, meta, "perspective")
    del res
    out = ap.apply_motion(api.hm._normalize_video_input(frames), meta, (127, 127, 127), framing_mode="crop_and_pad",
                          interpolation="bicubic", motion_blur=0.5, motion_blur_samples=17, ctx=ctx, keep_on_device=True)
    m64 = np.array([e["matrix"] for e in meta["motion_meta"]["per_frame"]], np.float64)
    for i in (0, 1, 127, 254, 255):
        lo = min(i, n - 2)
        ref, ref_mask = oracle.warp_blur_clip(host[lo:lo + 2], m64[lo:lo + 2], (w, h), 0.5, 17, interp="bicubic", border=BORDER)
        k = i - lo
        assert np.array_equal(out.frames[i].cpu().numpy(), ref[k]), i
        assert np.array_equal(out.masks[i, ..., 0].cpu().numpy(), ref_mask[k]), i


def test_c5_at_per_gpu_share_sampled_frames_match_oracle(api, ctx, oracle):
    """BASELINE configs[4] at the size bench.py times on one GPU (`--workload c5`, and the `motion_apply` leg): one GPU's
    share of the 8-GPU run = 64 x 3840x2160, Flow similarity + expand -> Motion Apply (expand, bilinear, blur 0.5,
    Ultra = 33 samples) on the original frames, device-resident.  All 63 pairs of the Flow stage against the oracle; the
    Flow node's warped frames {0, 31, 63} and masks bit-exact against the oracle warp of the reported matrices; the
    blurred frames {0, 1, 31, 62, 63} bit-exact from 2-frame oracle windows on the expanded canvas (a blurred frame needs
    its own pixels, its own and its successor's matrix -- the last frame its predecessor's: motion_apply.py:125-134; the
    canvas and its shift come from ALL the clip's matrices: motion_apply.py:288-294)."""
    import torch

    import bench
    from vstab_amd import apply_pipeline as ap

    n, h, w = 64, 2160, 3840
    frames = bench.synth_clip(n, 0, h, w, torch.device("cuda"))
    res = api.fp._stabilize_frames(api.hm._normalize_video_input(frames), *bench.C5_FLOW_ARGS, ctx=ctx, keep_on_device=True)
    meta = res.meta
    ow, oh = meta["framing"]["expanded_size"]
    assert tuple(res.frames.shape) == (n, oh, ow, 3) and ow > w and oh > h
    host = frames.cpu().numpy()
    check_flow_stage(api, oracle, host, meta, "similarity")
    fm = np.array([e["applied_matrix"] for e in meta["stabilization_warp"]["per_frame"]], np.float32)
    for i in (0, 31, 63):
        ref, ref_mask, cnt = oracle.warp_clip(host[i:i + 1], fm[i:i + 1], (ow, oh), border=BORDER)
        assert np.array_equal(res.frames[i].cpu().numpy(), ref[0]), i
        assert np.array_equal(res.masks[i, ..., 0].cpu().numpy(), ref_mask[0]), i
    del res
    torch.cuda.empty_cache()
    out = ap.apply_motion(api.hm._normalize_video_input(frames), meta, (127, 127, 127), ctx=ctx, keep_on_device=True, **bench.C5_APPLY)
    ma = out.meta["motion_apply"]
    ew, eh = ma["output_size"]
    assert ma["framing_mode"] == "expand" and ma["motion_blur_samples"] == 33 and tuple(out.frames.shape) == (n, eh, ew, 3)
    m64 = [np.array(e["matrix"], np.float64) for e in meta["motion_meta"]["per_frame"]]
    mins, maxs = api.hm._compute_bounding_boxes(m64, w, h)
    shift, size = api.hm._prepare_expand_transform(mins, maxs)
    assert list(size) == [ew, eh]
    expanded = np.stack([shift @ m for m in m64])
    for i in (0, 1, 31, 62, 63):
        lo = min(i, n - 2)
        ref, ref_mask = oracle.warp_blur_clip(host[lo:lo + 2], expanded[lo:lo + 2], (ew, eh), 0.5, 33, interp="bilinear", border=BORDER)
        k = i - lo
        assert np.array_equal(out.frames[i].cpu().numpy(), ref[k]), i
        assert np.array_equal(out.masks[i, ..., 0].cpu().numpy(), ref_mask[k]), i
