"""GPU parity: vstab_warp_batch / vstab_warp_blur_batch (HIP, via the C ABI) vs oracle/vo_warp.c.

Tolerance: bit-exact (np.array_equal) -- both sides evaluate the same f64 coordinate math and the
same left-to-right f32 sums with FMA contraction disabled."""

import numpy as np
import pytest

from tests.util import synth_frames, test_matrices as make_matrices

pytestmark = pytest.mark.gpu

BORDER = (np.array([127, 127, 127], np.float32) / 255.0)

CASES = [
    # (n, src_h, src_w, out_h, out_w, kind)
    (3, 24, 32, 24, 32, "identity"),
    (3, 24, 32, 24, 32, "translation"),
    (2, 45, 73, 45, 73, "similarity"),
    (2, 45, 73, 51, 80, "perspective"),      # odd sizes -> scalar store path
    (2, 120, 212, 120, 212, "similarity"),   # > 1 tile, vector store path
    (2, 120, 212, 130, 224, "perspective"),
    (1, 24, 32, 24, 32, "far"),
    (2, 10, 150, 10, 150, "similarity"),     # dh < 16 -> 102-wide OpenCV column blocks
]


@pytest.mark.parametrize("interp", ["bilinear", "bicubic"])
@pytest.mark.parametrize("case", CASES)
def test_warp_matches_oracle(ctx, oracle, case, interp):
    n, sh, sw, dh, dw, kind = case
    frames = synth_frames(n, sh, sw, seed=n + sh)
    mats = make_matrices(n, sw, sh, kind).astype(np.float32)
    ref, ref_mask, ref_cnt = oracle.warp_clip(frames, mats, (dw, dh), interp=interp, border=BORDER)
    dst, mask, cnt = ctx.warp_batch(frames, mats, (dw, dh), interp=interp, border=BORDER, subpix="q5",
                                    want_mask=True, want_count=True)
    assert np.array_equal(dst.cpu().numpy(), ref)
    assert np.array_equal(mask.cpu().numpy(), ref_mask)
    assert np.array_equal(cnt.cpu().numpy().astype(np.uint32), ref_cnt)


@pytest.mark.parametrize("case", CASES[:6])
def test_warp_exact_mode_matches_oracle(ctx, oracle, case):
    n, sh, sw, dh, dw, kind = case
    frames = synth_frames(n, sh, sw, seed=7)
    mats = make_matrices(n, sw, sh, kind).astype(np.float32)
    ref, ref_mask, _ = oracle.warp_clip(frames, mats, (dw, dh), interp="bilinear", border=BORDER, subpix="exact")
    dst, mask, _ = ctx.warp_batch(frames, mats, (dw, dh), interp="bilinear", border=BORDER, subpix="exact")
    assert np.array_equal(dst.cpu().numpy(), ref)
    assert np.array_equal(mask.cpu().numpy(), ref_mask)


STRESS = ["horizon", "flip", "minify", "magnify", "quarter_turn"]


@pytest.mark.parametrize("interp", ["bilinear", "bicubic"])
@pytest.mark.parametrize("kind", STRESS)
def test_warp_stress_matrices_match_oracle(ctx, oracle, kind, interp):
    """Maps whose inverse leaves the comfortable range: W == 0 and sign changes (INT clamp, short saturation), mirrored
    and rotated sources, 40x minification and 100x magnification (every 1/32-px phase of one source pixel)."""
    n, sh, sw, dh, dw = 3, 45, 73, 51, 80
    frames = synth_frames(n, sh, sw, seed=11)
    mats = make_matrices(n, sw, sh, kind).astype(np.float32)
    ref, ref_mask, ref_cnt = oracle.warp_clip(frames, mats, (dw, dh), interp=interp, border=BORDER)
    dst, mask, cnt = ctx.warp_batch(frames, mats, (dw, dh), interp=interp, border=BORDER, subpix="q5", want_mask=True, want_count=True)
    assert np.array_equal(dst.cpu().numpy(), ref, equal_nan=True)
    assert np.array_equal(mask.cpu().numpy(), ref_mask)
    assert np.array_equal(cnt.cpu().numpy().astype(np.uint32), ref_cnt)
    if interp == "bilinear" and kind != "horizon":
        ref, ref_mask, _ = oracle.warp_clip(frames, mats, (dw, dh), interp="bilinear", border=BORDER, subpix="exact")
        dst, mask, _ = ctx.warp_batch(frames, mats, (dw, dh), interp="bilinear", border=BORDER, subpix="exact")
        assert np.array_equal(dst.cpu().numpy(), ref, equal_nan=True)
        assert np.array_equal(mask.cpu().numpy(), ref_mask)


@pytest.mark.parametrize("kind", ["horizon", "flip", "magnify"])
def test_blur_stress_matrices_match_oracle(ctx, oracle, kind):
    frames = synth_frames(3, 45, 73, seed=13)
    mats = make_matrices(3, 73, 45, kind)
    ref, ref_mask = oracle.warp_blur_clip(frames, mats, (73, 45), 0.7, 5, interp="bilinear", border=BORDER)
    dst, mask = ctx.warp_blur_batch(frames, mats, (73, 45), 0.7, 5, interp="bilinear", border=BORDER)
    assert np.array_equal(dst.cpu().numpy(), ref, equal_nan=True)
    assert np.array_equal(mask.cpu().numpy(), ref_mask)


def test_warp_masks_zero_path(ctx, oracle):
    frames = synth_frames(2, 45, 73, seed=3)
    mats = make_matrices(2, 73, 45, "similarity").astype(np.float32)
    ref, _, _ = oracle.warp_clip(frames, mats, (73, 45), border=BORDER, want_mask=False)
    dst, mask, cnt = ctx.warp_batch(frames, mats, (73, 45), border=BORDER, want_mask=False)
    assert mask is None and cnt is None
    assert np.array_equal(dst.cpu().numpy(), ref)


@pytest.mark.parametrize("interp,samples", [("bilinear", 5), ("bicubic", 17), ("bilinear", 33)])
@pytest.mark.parametrize("kind", ["similarity", "perspective"])
def test_blur_matches_oracle(ctx, oracle, interp, samples, kind):
    n, sh, sw = 3, 45, 73
    frames = synth_frames(n, sh, sw, seed=11)
    mats = make_matrices(n, sw, sh, kind)
    ref, ref_mask = oracle.warp_blur_clip(frames, mats, (sw, sh), 0.5, samples, interp=interp, border=BORDER)
    dst, mask = ctx.warp_blur_batch(frames, mats, (sw, sh), 0.5, samples, interp=interp, border=BORDER)
    assert np.array_equal(dst.cpu().numpy(), ref)
    assert np.array_equal(mask.cpu().numpy(), ref_mask)


def test_blur_single_frame_quirk(ctx, oracle):
    """motion_apply.py:125-127,195: a 1-frame clip yields one sample but is still divided by S."""
    frames = synth_frames(1, 24, 32, seed=5)
    mats = make_matrices(1, 32, 24, "translation")
    ref, ref_mask = oracle.warp_blur_clip(frames, mats, (32, 24), 0.5, 9, border=BORDER)
    dst, mask = ctx.warp_blur_batch(frames, mats, (32, 24), 0.5, 9, border=BORDER)
    assert np.array_equal(dst.cpu().numpy(), ref)
    assert np.array_equal(mask.cpu().numpy(), ref_mask)


def test_identity_is_passthrough(ctx):
    """KA1 (check_motion_meta.py:289-311): identity -> output == input within 1e-6, mask all zero."""
    frames = synth_frames(3, 24, 32, seed=2)
    mats = np.tile(np.eye(3, dtype=np.float32), (3, 1, 1))
    dst, mask, cnt = ctx.warp_batch(frames, mats, (32, 24), border=BORDER, want_count=True)
    assert np.max(np.abs(dst.cpu().numpy() - frames)) <= 1e-6
    assert float(mask.max()) == 0.0 and int(cnt.sum()) == 0


def test_full_size_properties(ctx):
    """1080p (BASELINE configs[1] frame size): identity passthrough + pure integer translation
    equals an array shift, independent of the oracle."""
    import torch

    n, h, w = 2, 1080, 1920
    g = torch.Generator(device="cpu").manual_seed(0)
    frames = torch.rand((n, h, w, 3), generator=g, dtype=torch.float32)
    eye = np.tile(np.eye(3, dtype=np.float32), (n, 1, 1))
    dst, mask, cnt = ctx.warp_batch(frames, eye, (w, h), border=BORDER, want_count=True)
    assert torch.equal(dst.cpu(), frames)
    assert int(cnt.sum()) == 0
    shift = eye.copy()
    shift[:, 0, 2] = 5.0
    shift[:, 1, 2] = -3.0
    dst, mask, cnt = ctx.warp_batch(frames, shift, (w, h), border=BORDER, want_count=True)
    d = dst.cpu()
    assert torch.equal(d[:, : h - 3, 5:, :], frames[:, 3:, : w - 5, :])
    m = mask.cpu()
    assert float(m[:, : h - 3, 5:].max()) == 0.0
    assert float(m[:, h - 3 :, :].min()) == 1.0 and float(m[:, :, :5].min()) == 1.0
    expect = h * w - (h - 3) * (w - 5)
    assert cnt.cpu().tolist() == [expect] * n


def test_c_abi_error_reporting(ctx):
    """Failures cross the ABI as a status + vstab_last_error(), surfaced as VstabError with the message."""
    import torch

    from vstab_amd import native

    frames = torch.zeros((1, 8, 8, 3))
    with pytest.raises(native.VstabError, match="unknown interpolation|interp"):
        ctx.lib.vstab_warp_batch.argtypes  # noqa: B018 - signature is set
        native._check(ctx.lib.vstab_warp_batch(ctx.handle, frames.cuda().data_ptr(), 1, 8, 8, np.eye(3, dtype=np.float32).ctypes.data,
                                               8, 8, 7, BORDER.ctypes.data, 0, frames.cuda().data_ptr(), None, None), "vstab_warp_batch")
    with pytest.raises(native.VstabError, match="exact sub-pixel mode exists for bilinear only"):
        ctx.warp_batch(frames, np.eye(3, dtype=np.float32)[None], (8, 8), interp="bicubic", subpix="exact")
    with pytest.raises(native.VstabError, match="samples=40"):
        native._check(ctx.lib.vstab_warp_blur_batch(ctx.handle, frames.cuda().data_ptr(), 1, 8, 8, np.eye(3).ctypes.data,
                                                    np.zeros(40).ctypes.data, 40, 8, 8, 0, BORDER.ctypes.data, 0,
                                                    frames.cuda().data_ptr(), None), "vstab_warp_blur_batch")
    with pytest.raises(native.VstabError, match="at least 2 frames|at least two"):
        ctx.dis_flow_batch(torch.zeros((1, 64, 64), dtype=torch.uint8).cuda())
    with pytest.raises(native.VstabError, match="too small for DIS"):
        ctx.dis_flow_batch(torch.zeros((2, 6, 6), dtype=torch.uint8).cuda())


def test_every_entry_point_rejects_empty_arguments(pkg):
    """A live context with every other argument NULL / zero: each compute entry point returns a non-zero status with a
    message before it launches anything, and the context is still usable afterwards."""
    import ctypes

    import torch

    from tests.test_abi_cpu import header_prototypes
    from vstab_amd import native

    c = native.Context(0)
    lib = c.lib
    # calls that legitimately succeed with these arguments (zero-byte transfers are no-ops)
    benign = {"vstab_destroy", "vstab_create", "vstab_set_stream", "vstab_synchronize", "vstab_set_timing", "vstab_dis_set_clip_start",
              "vstab_upload", "vstab_download", "vstab_upload_f32_coded", "vstab_upload_u8_as_f32", "vstab_download_mask_coded"}
    checked = 0
    for name, params in header_prototypes():
        if not params or not params[0].startswith("vstab_ctx*") or name in benign:
            continue
        argtypes, args = [ctypes.c_void_p], [c.handle]
        for ptxt in params[1:]:
            if "*" in ptxt:
                argtypes.append(ctypes.c_void_p); args.append(None)
            elif ptxt.startswith(("double", "const double")):
                argtypes.append(ctypes.c_double); args.append(0.0)
            elif ptxt.startswith(("float", "const float")):
                argtypes.append(ctypes.c_float); args.append(0.0)
            elif ptxt.startswith("size_t"):
                argtypes.append(ctypes.c_size_t); args.append(0)
            else:
                argtypes.append(ctypes.c_int); args.append(0)
        fn = getattr(lib, name)
        saved = (fn.argtypes, fn.restype)
        fn.argtypes, fn.restype = argtypes, ctypes.c_int
        try:
            rc = fn(*args)
        finally:
            fn.argtypes, fn.restype = saved
        msg = lib.vstab_last_error()
        assert rc != 0 and msg, f"{name}(ctx, zeros) returned {rc} / {msg!r}"
        checked += 1
    assert checked >= 16
    frames = torch.rand((1, 16, 24, 3))
    dst, _, _ = c.warp_batch(frames, np.eye(3, dtype=np.float32)[None], (24, 16), border=BORDER)
    assert torch.equal(dst.cpu(), frames)
    c.close()


def test_odd_output_sizes_and_single_pixel_frames(ctx, oracle):
    """Ragged shapes: 1-pixel-wide / 1-row sources, outputs not divisible by 4, output larger than source."""
    rng = np.random.default_rng(0)
    for (sh, sw, dh, dw) in [(1, 9, 3, 11), (7, 1, 9, 5), (5, 5, 17, 19), (33, 65, 31, 63)]:
        frames = rng.random((2, sh, sw, 3), dtype=np.float32)
        mats = np.tile(np.eye(3, dtype=np.float32), (2, 1, 1))
        mats[:, 0, 2] = [0.4, -1.3]
        mats[:, 1, 2] = [1.6, 0.2]
        for interp in ("bilinear", "bicubic"):
            ref, ref_mask, ref_cnt = oracle.warp_clip(frames, mats, (dw, dh), interp=interp, border=BORDER)
            dst, mask, cnt = ctx.warp_batch(frames, mats, (dw, dh), interp=interp, border=BORDER, want_count=True)
            assert np.array_equal(dst.cpu().numpy(), ref) and np.array_equal(mask.cpu().numpy(), ref_mask)
            assert np.array_equal(cnt.cpu().numpy().astype(np.uint32), ref_cnt)


@pytest.mark.parametrize("nbytes", [4, 1000, (4 << 20) - 4, (32 << 20), (32 << 20) * 5 + 12, (32 << 20) * 9 - 8])
def test_pipelined_upload_download_round_trip(ctx, nbytes):
    """vstab_upload / vstab_download (the node boundary's transfers through the pinned ring + host thread team): byte
    exact for sizes below one ring slot, equal to it, wrapping the ring several times and ragged tails; a download
    enqueued right after a kernel on the same stream sees its result."""
    import torch

    n = nbytes // 4
    g = torch.Generator().manual_seed(nbytes % 1000)
    host = torch.randint(-2 ** 31, 2 ** 31 - 1, (n,), generator=g, dtype=torch.int32)
    dev = ctx.upload(host)
    assert dev.is_cuda and torch.equal(dev.cpu(), host)
    back = ctx.download(dev)
    assert back.device.type == "cpu" and not back.is_pinned() and torch.equal(back, host)
    bumped = dev + 1                       # stream-ordered producer, no synchronisation before the download
    assert torch.equal(ctx.download(bumped), host + 1)
    again = ctx.upload(host[: max(1, n // 3)])   # the ring is reused by a later call while nothing else synchronised
    assert torch.equal(again.cpu(), host[: max(1, n // 3)])



@pytest.mark.parametrize("interp,samples", [("bilinear", 9), ("bicubic", 5), ("bicubic", 17), ("bilinear", 33)])
@pytest.mark.parametrize("src,out,kind", [((270, 480), (270, 480), "similarity"),      # interior blocks + a border ring
                                          ((270, 480), (300, 523), "similarity"),      # ragged last tile column / row
                                          ((200, 333), (200, 333), "perspective"),     # perspective samples: general loop only
                                          ((160, 400), (12, 400), "translation"),      # dh < 16: OpenCV column blocks of 85 px, not tile-aligned
                                          ((128, 128), (128, 128), "identity")])
def test_blur_interior_fast_path_matches_oracle(ctx, oracle, monkeypatch, interp, samples, src, out, kind):
    """warp_blur_kernel classifies each tile once (all four corners, every sample, interior with margin) and runs the
    interior tiles through a loop without bounds tests, with the weight PRODUCTS from the LDS table and block-origin
    terms shared by a thread's two pixels; everything else takes the general loop.  Both must give the oracle's bits,
    and the same bits as the general loop forced everywhere (VSTAB_BLUR_FAST=0)."""
    n = 3
    sh, sw = src
    dh, dw = out
    frames = synth_frames(n, sh, sw, seed=sh + samples)
    mats = make_matrices(n, sw, sh, kind, seed=7)
    ref, ref_mask = oracle.warp_blur_clip(frames, mats, (dw, dh), 0.5, samples, interp=interp, border=BORDER)
    dst, mask = ctx.warp_blur_batch(frames, mats, (dw, dh), 0.5, samples, interp=interp, border=BORDER)
    assert np.array_equal(dst.cpu().numpy(), ref) and np.array_equal(mask.cpu().numpy(), ref_mask)
    monkeypatch.setenv("VSTAB_BLUR_FAST", "0")
    dst0, mask0 = ctx.warp_blur_batch(frames, mats, (dw, dh), 0.5, samples, interp=interp, border=BORDER)
    assert np.array_equal(dst0.cpu().numpy(), ref) and np.array_equal(mask0.cpu().numpy(), ref_mask)
    # no mask requested: same pixels
    dst1, mask1 = ctx.warp_blur_batch(frames, mats, (dw, dh), 0.5, samples, interp=interp, border=BORDER, want_mask=False)
    assert mask1 is None and np.array_equal(dst1.cpu().numpy(), ref)


@pytest.mark.parametrize("interp", ["bilinear", "bicubic"])
def test_blur_single_frame_quirk_on_the_staged_path(ctx, oracle, interp):
    """motion_apply.py:125-127,195 on an interior tile: a 1-frame clip yields ONE sample matrix but is still divided by S,
    for the pixels and for the coverage alike (mask = 1 - 1/S) -- the staged-window loop must keep that quirk."""
    frames = synth_frames(1, 160, 200, seed=8)
    mats = make_matrices(1, 200, 160, "similarity", seed=2)
    mats[0, :2, 2] *= 0.2
    ref, ref_mask = oracle.warp_blur_clip(frames, mats, (200, 160), 0.5, 9, interp=interp, border=BORDER)
    dst, mask = ctx.warp_blur_batch(frames, mats, (200, 160), 0.5, 9, interp=interp, border=BORDER)
    assert np.array_equal(dst.cpu().numpy(), ref) and np.array_equal(mask.cpu().numpy(), ref_mask)
    assert abs(float(ref_mask[0, 80, 100]) - (1.0 - 1.0 / 9.0)) < 1e-6


@pytest.mark.parametrize("interp", ["bilinear", "bicubic"])
@pytest.mark.parametrize("kind", ["flip", "quarter_turn", "magnify", "minify", "horizon", "far"])
def test_blur_staged_path_under_extreme_maps(ctx, oracle, kind, interp):
    """The tile classification of warp_blur_kernel under maps that stress its assumptions: mirrored and rotated axes
    (the per-pixel evaluation is monotone DEcreasing: the corners still bound it), a 37-137x zoom (a source window of a
    few texels), a 20-50x reduction (a window far too large to stage: general loop), a projective horizon inside the frame
    (non-affine samples: general loop) and a map that leaves the source entirely.  Bit-exact against the oracle."""
    n, sh, sw = 3, 270, 480
    frames = synth_frames(n, sh, sw, seed=17)
    mats = make_matrices(n, sw, sh, kind, seed=3)
    ref, ref_mask = oracle.warp_blur_clip(frames, mats, (sw, sh), 0.6, 9, interp=interp, border=BORDER)
    dst, mask = ctx.warp_blur_batch(frames, mats, (sw, sh), 0.6, 9, interp=interp, border=BORDER)
    assert np.array_equal(dst.cpu().numpy(), ref, equal_nan=True) and np.array_equal(mask.cpu().numpy(), ref_mask)


@pytest.mark.parametrize("kind,samples", [("similarity", 9), ("translation", 33), ("similarity", 5)])
def test_blur_exact_subpixel_mode_staged_path_matches_oracle(ctx, oracle, kind, samples):
    """`subpix="exact"` (the OpenCV >= 4.11 bilinear warp: float32 coordinates, no 1/32-px quantisation) in the S-sample
    blur: since round 4 its tiles are staged in LDS like the default sampler's (border-capable loop), instead of falling to
    the general loop at 4K / S = 33.  Bit-exact against the oracle's exact mode, content edges and padded regions included;
    and the same bits as the general loop forced everywhere."""
    import os

    h, w = 270, 480
    frames = synth_frames(3, h, w, seed=21)
    mats = make_matrices(3, w, h, kind, seed=4)
    ref, ref_mask = oracle.warp_blur_clip(frames, mats, (w, h), 0.5, samples, interp="bilinear", border=BORDER, subpix="exact")
    dst, mask = ctx.warp_blur_batch(frames, mats, (w, h), 0.5, samples, interp="bilinear", border=BORDER, subpix="exact")
    assert np.array_equal(dst.cpu().numpy(), ref) and np.array_equal(mask.cpu().numpy(), ref_mask)
    os.environ["VSTAB_BLUR_FAST"] = "0"
    try:
        gen, gen_mask = ctx.warp_blur_batch(frames, mats, (w, h), 0.5, samples, interp="bilinear", border=BORDER, subpix="exact")
    finally:
        del os.environ["VSTAB_BLUR_FAST"]
    assert bool((gen == dst).all()) and bool((gen_mask == mask).all())


def test_pad_counts_reach_the_host_behind_the_warp(ctx):
    """The plain warp's per-frame padded-pixel counts are mirrored into coherent host memory by a one-workgroup kernel behind
    the warp; `last_pad_counts` waits for that word alone.  Same numbers as the device tensor, for calls of different
    frame counts in a row, and the handle refuses a stale frame count."""
    import torch

    for n, seed in ((5, 1), (1, 2), (33, 3)):
        frames = torch.from_numpy(synth_frames(n, 72, 128, seed=seed)).cuda()
        mats = make_matrices(n, 128, 72, "similarity", seed=seed).astype(np.float32)
        _, mask, counts = ctx.warp_batch(frames, mats, (128, 72), border=(0.5, 0.5, 0.5), want_mask=True, want_count=True)
        host = counts._vstab_fetch()
        assert np.array_equal(host, counts.cpu().numpy().astype(np.int64))
        assert np.array_equal(host, (mask.cpu().numpy() > 0).reshape(n, -1).sum(axis=1))
    with pytest.raises(Exception, match="no warp with counts over 7 frames"):
        ctx.last_pad_counts(7)
