"""The CPU suite's share of the oracle-independent evidence (the GPU half: tests/test_analytic_gpu.py).

The oracle's estimation chain (gray -> DIS -> stride-8 sampling -> model fit: the restatement of cv2.cvtColor / resize,
cv2.DISOpticalFlow, cv2.estimateAffinePartial2D / findHomography that nothing reference-held can pin) must recover the
known motion of analytic clips -- clips whose every pixel is a closed-form function of a known camera path, so the true
transition of every pair is known exactly.  The HIP path equals the oracle bit for bit (tests/test_dis_gpu.py,
tests/test_fit_gpu.py), so what is shown here for the checker holds for the product.  The clips are 960x540 (the working size of every BASELINE config; bench.synth_clip scales its texture with the frame
width, so smaller frames would be a harder clip, not a faster test); bounds as in the GPU file."""

import numpy as np
import pytest

from tests.util import shake_path


def _clip(n, w, h, kind, amp, seed=3):
    import torch

    import bench

    cam = shake_path(n, w, h, kind, seed=seed, amp=amp)
    frames = bench.synth_clip(n, 0, h, w, torch.device("cpu"), mats=cam).numpy()
    return cam, frames


def test_transition_accuracy_metric_is_zero_for_the_truth_and_scales_with_resolution():
    import bench

    cam = shake_path(6, 1920, 1080, "perspective", amp=2.0)
    truth = [cam[i + 1] @ np.linalg.inv(cam[i]) for i in range(5)]
    acc = bench.transition_accuracy([m / m[2, 2] for m in truth], cam, (1920, 1080), (960, 540))
    assert acc["centre_px"]["max"] < 1e-9 and acc["corner_px"]["max"] < 1e-9 and acc["lin_2x2"]["max"] < 1e-12
    off = [np.array([[1, 0, 0.2], [0, 1, 0], [0, 0, 1.0]]) @ m for m in truth]     # 0.2 full-res px = 0.1 working px
    acc = bench.transition_accuracy(off, cam, (1920, 1080), (960, 540))
    assert abs(acc["centre_px"]["max"] - 0.1) < 1e-6 and abs(acc["corner_px"]["mean"] - 0.1) < 1e-3


@pytest.mark.parametrize("kind,amp,centre,corner,lin", [("translation", 1.0, 0.05, 0.05, 1e-12), ("similarity", 1.0, 0.05, 0.09, 2e-4),
                                                        ("perspective", 1.0, 0.06, 0.2, 1.0), ("similarity", 3.0, 0.05, 0.09, 2e-4)])
def test_oracle_recovers_known_camera_motion(oracle, pkg, kind, amp, centre, corner, lin):
    import bench
    from vstab_amd import flow_pipeline as fp

    w, h, n = 960, 540, 4
    cam, frames = _clip(n, w, h, kind, amp)
    gray = oracle.gray_for_estimation(frames, None)
    flow = oracle.dis_flow_clip(gray)
    recs = [oracle.fit_all_modes(flow[i], 8, kind)[0] for i in range(n - 1)]
    mats, modes, _, _, _ = fp.select_transitions(recs, kind)
    assert modes == [kind] * (n - 1)
    acc = bench.transition_accuracy(mats, cam, (w, h), None)
    assert acc["true_motion_px"]["max"] > 0.5
    assert acc["centre_px"]["max"] <= centre and acc["corner_px"]["max"] <= corner and acc["lin_2x2"]["max"] <= lin, acc


def test_homography_fit_alone_is_exact_on_an_analytic_flow_field(oracle):
    """The fit without DIS: a flow field computed in closed form from a known homography (plus bounded noise) must give
    that homography back -- 1e-9 px without noise, a few thousandths of a pixel with 0.05 px of noise."""
    h, w = 270, 480
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    H = np.array([[1.002, -0.003, 3.2], [0.003, 1.002, -1.7], [4e-6, -2e-6, 1.0]])
    den = H[2, 0] * xx + H[2, 1] * yy + 1.0
    exact = np.stack([(H[0, 0] * xx + H[0, 1] * yy + H[0, 2]) / den - xx, (H[1, 0] * xx + H[1, 1] * yy + H[1, 2]) / den - yy], -1)
    pts = np.array([[0, 0, 1.0], [w - 1, 0, 1], [0, h - 1, 1], [w - 1, h - 1, 1], [w / 2, h / 2, 1]]).T
    want = (H @ pts)[:2] / (H @ pts)[2]
    for noise, bound in ((0.0, 1e-4), (0.05, 0.02)):
        flow = (exact + np.random.default_rng(1).normal(0, noise, exact.shape)).astype(np.float32)
        out, _, _ = oracle.fit_all_modes(flow, 8, "perspective")
        m = out["perspective"]["matrix"].astype(np.float64)
        got = (m @ pts)[:2] / (m @ pts)[2]
        assert out["perspective"]["accepted"] and np.hypot(*(got - want)).max() < bound, (noise, np.hypot(*(got - want)).max())


@pytest.mark.parametrize("interp,bound", [("bilinear", 2e-2), ("bicubic", 2e-2)])
def test_oracle_warp_reproduces_an_analytic_texture(oracle, interp, bound):
    """warpPerspective's conventions without OpenCV: warping frame(p) = T(p) by M must give T(M^-1 p) up to the
    interpolation error of a band-limited texture and the 1/32-px coordinate quantisation (measured at this 480-px width,
    where the texture is 4x finer than at 1920: max 0.012, mean 0.003) -- the same comparison against a truth shifted by
    half a pixel gives 0.09, a transposed / inverted matrix far more.  M has sub-pixel translation, rotation, zoom and perspective."""
    import torch

    import bench

    w, h = 480, 270
    eye = np.eye(3)[None]
    src = bench.synth_clip(1, 0, h, w, torch.device("cpu"), mats=eye, seed=7).numpy()       # T itself (texture scaled to 1920-wide frequencies x4)
    m = np.array([[1.01 * np.cos(0.02), -1.01 * np.sin(0.02), 3.37], [1.01 * np.sin(0.02), 1.01 * np.cos(0.02), -2.61], [2e-5, -1e-5, 1.0]])
    want = bench.synth_clip(1, 0, h, w, torch.device("cpu"), mats=m[None], seed=7).numpy()[0]   # T(M^-1 p), sampled analytically
    got, cov = oracle.warp_frame(src[0], m.astype(np.float32), (w, h), interp=interp, border=(0.5, 0.5, 0.5))
    inside = cov > 0.5
    for _ in range(4):   # 4 px in from the covered region's rim: no tap of the kernel reaches the border colour
        inside = inside & np.roll(inside, 1, 0) & np.roll(inside, -1, 0) & np.roll(inside, 1, 1) & np.roll(inside, -1, 1)
    inside[:4] = inside[-4:] = False
    inside[:, :4] = inside[:, -4:] = False
    err = np.abs(got - want).max(axis=-1)
    assert inside.mean() > 0.85 and err[inside].max() < bound and err[inside].mean() < bound / 4, (err[inside].max(), err[inside].mean())


def test_oracle_bilinear_warp_against_scipy_map_coordinates(oracle):
    """An independent implementation as referee: scipy.ndimage.map_coordinates(order=1) samples the source at the
    inverse-mapped position of every output pixel (pixel centres at integers, as cv2.warpPerspective has them).  The
    oracle's unquantised bilinear path (`subpix="exact"`) must agree to float32 rounding of the coordinates (2e-4 on a
    noise image whose neighbouring pixels differ by up to 1.0), its default 1/32-px path to the quantisation bound
    (|gradient| <= 1 per px x 1/64 px per axis, two axes, bilinear: 0.04); pixels whose taps touch the border are compared
    too (constant border, per tap)."""
    from scipy import ndimage

    rng = np.random.default_rng(4)
    h, w = 90, 120
    src = rng.random((h, w, 3), dtype=np.float32)
    m = np.array([[1.03 * np.cos(0.05), -1.03 * np.sin(0.05), 4.3], [1.03 * np.sin(0.05), 1.03 * np.cos(0.05), -3.7], [1.5e-4, -1e-4, 1.0]])
    inv = np.linalg.inv(m.astype(np.float32).astype(np.float64))
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    den = inv[2, 0] * xx + inv[2, 1] * yy + inv[2, 2]
    sx = (inv[0, 0] * xx + inv[0, 1] * yy + inv[0, 2]) / den
    sy = (inv[1, 0] * xx + inv[1, 1] * yy + inv[1, 2]) / den
    border = (0.25, 0.5, 0.75)
    want = np.stack([ndimage.map_coordinates(src[..., c].astype(np.float64), [sy, sx], order=1, mode="constant", cval=border[c])
                     for c in range(3)], axis=-1)
    # scipy treats everything beyond the last pixel centre as `cval` without blending; OpenCV blends the border colour in
    # per tap.  Compare where all four taps are inside, and separately check that far-outside pixels are pure border.
    inside = (sx >= 0) & (sx <= w - 1) & (sy >= 0) & (sy <= h - 1)
    exact, _ = oracle.warp_frame(src, m.astype(np.float32), (w, h), interp="bilinear", border=border, subpix="exact")
    q5, _ = oracle.warp_frame(src, m.astype(np.float32), (w, h), interp="bilinear", border=border, subpix="q5")
    assert inside.mean() > 0.8
    assert np.abs(exact - want)[inside].max() < 2e-4
    assert np.abs(q5 - want)[inside].max() < 0.04 and np.abs(q5 - want)[inside].mean() < 0.008
    far = (sx < -2) | (sx > w + 1) | (sy < -2) | (sy > h + 1)
    assert far.any() and np.array_equal(q5[far], np.broadcast_to(np.float32(border), q5.shape)[far])
