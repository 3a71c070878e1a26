"""GPU parity: vstab_gray_downscale vs oracle/vo_gray.c -- bit-exact (u8)."""

import numpy as np
import pytest

from tests.util import synth_frames

pytestmark = pytest.mark.gpu

CASES = [
    # (n, h, w, work (w,h) or None)
    (2, 48, 64, None),          # no downscale (<= 960 px): gray only
    (2, 45, 73, None),          # odd width: scalar tail of the gray conversion, unaligned stores
    (2, 108, 192, (96, 54)),    # exact 2x (the 1080p -> 960x540 case in miniature)
    (2, 216, 384, (96, 54)),    # exact 4x (the 4K case in miniature)
    (2, 90, 150, (50, 30)),     # exact 3x -> generic integer path
    (2, 72, 128, (96, 54)),     # 1.333x -> general area path (720p -> 960x540 in miniature)
    (1, 67, 121, (60, 33)),     # ragged general ratio
]


@pytest.mark.parametrize("case", CASES)
def test_gray_downscale_matches_oracle(ctx, oracle, case):
    n, h, w, work = case
    frames = synth_frames(n, h, w, seed=h)
    ref = oracle.gray_for_estimation(frames, work)
    got = ctx.gray_downscale(frames, work).cpu().numpy()
    assert got.dtype == np.uint8 and got.shape == ref.shape
    assert np.array_equal(got, ref)


@pytest.mark.parametrize("case", CASES[:4] + [(2, 40, 1000, (500, 20)), (1, 16, 2044, (511, 4))])
def test_gray_out_of_range_samples_match_oracle(ctx, oracle, case):
    """Samples outside [0, 1] (overshoot of an upstream bicubic resize, HDR-ish sources, infinities): the clip to
    [0, 255] before the truncation decides; long rows = several passes of a workgroup over the row."""
    n, h, w, work = case
    rng = np.random.default_rng(w)
    frames = rng.uniform(-0.6, 1.8, (n, h, w, 3)).astype(np.float32)
    frames[0, 1, 2::7, 0] = np.inf
    frames[0, 2, 3::5, 2] = -np.inf
    frames[-1, 3, ::9, 1] = 1.0e30
    ref = oracle.gray_for_estimation(frames, work)
    got, peaks = ctx.gray_downscale(frames, work, want_range=True)
    assert np.array_equal(got.cpu().numpy(), ref)
    assert np.array_equal(peaks.cpu().numpy(), frames.reshape(n, -1).max(axis=1))


def test_gray_1080p_properties(ctx):
    """Full BASELINE size: constant frames -> constant gray (truncation of 255*Y), any ratio."""
    import torch

    frames = torch.full((2, 1080, 1920, 3), 0.5, dtype=torch.float32)
    g = ctx.gray_downscale(frames, (960, 540)).cpu().numpy()
    assert g.shape == (2, 540, 960) and np.all(g == 127)
    frames4k = torch.full((1, 2160, 3840, 3), 1.0, dtype=torch.float32)
    g = ctx.gray_downscale(frames4k, (960, 540)).cpu().numpy()
    assert np.all(g == 255) or np.all(g == 254)


def test_4k_sizes(ctx, oracle):
    """BASELINE configs[4] frame size (3840x2160): 4x4 area path vs the oracle on two frames, and an
    identity warp at 4K (size-independent property)."""
    import torch

    g = torch.Generator().manual_seed(4)
    frames = torch.rand((2, 2160, 3840, 3), generator=g, dtype=torch.float32)
    got = ctx.gray_downscale(frames, (960, 540)).cpu().numpy()
    ref = oracle.gray_for_estimation(frames.numpy(), (960, 540))
    assert np.array_equal(got, ref)
    eye = np.tile(np.eye(3, dtype=np.float32), (2, 1, 1))
    dst, mask, cnt = ctx.warp_batch(frames, eye, (3840, 2160), border=(0.5, 0.5, 0.5), want_count=True)
    assert torch.equal(dst.cpu(), frames) and int(cnt.sum()) == 0


@pytest.mark.parametrize("case", CASES)
def test_range_sniff_from_the_gray_pass(ctx, case):
    """F0's per-frame `float(arr.max()) > 1.5` (stabilizer_utils.py:127-131) comes out of the gray pass: the per-frame
    maxima equal numpy's, exactly, for every resize path (fused 1x/2x/4x boxes, two-pass general ratios), and the gray
    image is the one the plain entry point produces.  NaN propagates as in numpy (the comparison is then False)."""
    n, h, w, work = case
    rng = np.random.default_rng(h * 7 + w)
    frames = synth_frames(n, h, w, seed=h)
    frames[0] *= 255.0                                   # a 0..255 float frame
    frames[0, h - 1, w - 1, 2] = 300.5                   # the maximum sits in the last sample of the frame
    if n > 1:
        frames[1, rng.integers(h), rng.integers(w), rng.integers(3)] = np.nan
    gray, peaks = ctx.gray_downscale(frames, work, want_range=True)
    assert np.array_equal(gray.cpu().numpy(), ctx.gray_downscale(frames, work).cpu().numpy())
    got = peaks.cpu().numpy()
    want = frames.reshape(n, -1).max(axis=1)
    assert got.dtype == np.float32 and np.array_equal(got, want, equal_nan=True)
    assert got[0] == np.float32(300.5) and (n == 1 or np.isnan(got[1]))
    alone = ctx.frame_range(frames).cpu().numpy()
    assert np.array_equal(alone, want, equal_nan=True)
    # the rule itself (vstab_apply_value_range): IEEE float32 division of the frames above 1.5, NaN maxima compare False
    expect = frames.copy()
    expect[0] = frames[0] / np.float32(255.0)
    rescaled = ctx.apply_value_range(frames, peaks).cpu().numpy()
    assert np.array_equal(rescaled, expect, equal_nan=True)


def test_frame_range_at_full_size(ctx):
    import torch

    g = torch.Generator().manual_seed(9)
    frames = torch.rand((3, 1080, 1920, 3), generator=g, dtype=torch.float32)
    frames[2, 1079, 1919, 1] = 1.75
    want = frames.reshape(3, -1).amax(dim=1)
    dev = frames.cuda()
    assert torch.equal(ctx.frame_range(dev).cpu(), want)
    _, peaks = ctx.gray_downscale(dev, (960, 540), want_range=True)
    assert torch.equal(peaks.cpu(), want)


def test_range_sniff_with_both_infinities(ctx, oracle):
    """a frame holding +inf and -inf has max +inf (numpy, stabilizer_utils.py:127-131: rescaled), not NaN
    -- also when the two sit in one pixel / one 16-byte load, where a sum-based NaN probe would see inf - inf."""
    frames = synth_frames(3, 270, 480, seed=2)
    frames[0, 10, 10] = (np.inf, -np.inf, 0.25)          # one pixel: the gray pass probes per pixel
    frames[1].reshape(-1)[8:12] = (np.inf, 0.5, -np.inf, 0.5)   # one aligned float4: the stand-alone pass probes per vector
    frames[2, 5, 5, 0] = -np.inf
    want = frames.reshape(3, -1).max(axis=1)
    assert np.isposinf(want[0]) and np.isposinf(want[1]) and np.isfinite(want[2])
    _, peaks = ctx.gray_downscale(frames, None, want_range=True)
    assert np.array_equal(peaks.cpu().numpy(), want)
    assert np.array_equal(ctx.frame_range(frames).cpu().numpy(), want)
    assert np.array_equal(oracle.frame_max(frames), want)


def test_frame_maxima_reach_the_host_without_a_copy(ctx):
    """The kernel that forms the per-frame maxima mirrors them into coherent host memory (its last workgroup: one store of
    all values, then a flag); `last_frame_peaks` waits for that kernel alone.  The host values are the device tensor's,
    NaN and infinities included, across passes of different frame counts (the pass's target count is cumulative), and
    `host_math.apply_value_range` takes them from there (`_vstab_fetch`) instead of a transfer of its own."""
    import torch
    from vstab_amd import host_math as hm

    for n, seed in ((3, 1), (1, 2), (40, 3), (3, 4)):
        frames = synth_frames(n, 90, 160, seed=seed)
        frames[0, 3, 4, 1] = 7.5
        if n > 2:
            frames[1, 5, 6, 2] = np.nan
            frames[2, 1, 1, 0] = -np.inf
        gray, peaks = ctx.gray_downscale(frames, None, want_range=True)
        host = ctx.last_frame_peaks(n)
        assert host.dtype == np.float32 and np.array_equal(host, peaks.cpu().numpy(), equal_nan=True)
        assert np.array_equal(host, frames.reshape(n, -1).max(axis=1), equal_nan=True)
        assert hasattr(peaks, "_vstab_fetch") and hm.prefetch_peaks(peaks) is peaks
        dev = torch.from_numpy(frames).cuda()
        rescaled, vrange = hm.apply_value_range(dev, peaks, ctx)
        want = frames.copy()
        want[0] /= np.float32(255.0)
        assert vrange == "0_255" and np.array_equal(rescaled.cpu().numpy(), want, equal_nan=True)
    with pytest.raises(Exception, match="no range pass over 5 frames"):
        ctx.last_frame_peaks(5)
