"""Oracle-independent accuracy of the Flow node (VERDICT r2 "next" #1c).

The OpenCV primitives cannot be pinned in this environment (no cv2, no reference-held vectors: DESIGN section 5), so
these tests are the evidence that separates "equals our restatement of DIS / RANSAC / warpPerspective" from "is right":
clips whose every pixel is an analytic function of a known camera path (bench.synth_clip: frame_i(p) = T(M_i^-1 p), no
interpolation anywhere), so the true transition M_{i+1} M_i^-1 of every pair and the image a locked camera must show
are known exactly.  Reference behaviour under test: video_stabilizer_flow.py:133-210 (estimation) and :356-371
(camera_lock / strength).

Bounds are at WORKING resolution (where DIS runs; 960x540 for both sizes) and were measured over 47 pairs per case with
tools/analytic_accuracy.py (profiles/r03_analytic_accuracy.md); each is the measured maximum plus ~30-50 % headroom:

  landscape, translation / similarity, motion up to ~10 px per frame
      displacement error at the frame centre  <= 0.05 px   (measured max 0.038 / 0.025)      SURVEY 8(c): 0.05 px
      displacement error at the worst corner  <= 0.075 px  (measured max 0.038 / 0.054)
      max |delta| of the 2x2 part             <= 1e-4      (measured max 6.2e-5)             SURVEY 8(c): 1e-4
  portrait 540x960: the clip's texture scales with the frame width (x 3.6 in frequency against 1920: 16 px wavelength),
      closer to what an 8 px patch can resolve -- a property of the clip, not of the estimator
      centre <= 0.13 px (0.093), corner <= 0.16 px (0.120), 2x2 <= 2.5e-4 (1.7e-4)
  perspective: see PERSPECTIVE_BOUNDS below (centre <= 0.08 px, worst corner <= 0.25 px: the homography has 8 degrees of
      freedom to spend on DIS's border bias; its 2x2 entries are not separately identifiable)

The previous gates (test_nodes_gpu.py: 0.8 px / 3e-3, translation mode 2.5 px / 2e-2) are replaced by these.
"""

import numpy as np
import pytest

from tests.util import shake_path

pytestmark = pytest.mark.gpu

N = 24

# (centre px, corner px, 2x2)
BOUNDS = {
    ("landscape", "translation"): (0.05, 0.075, 1e-12),
    ("landscape", "similarity"): (0.05, 0.075, 1e-4),
    ("portrait", "translation"): (0.13, 0.16, 1e-12),
    ("portrait", "similarity"): (0.13, 0.16, 2.5e-4),
}
# perspective: displacement only -- the entries of a homography's 2x2 part trade against its perspective row times the
# translation (H00 = a + tx * p0), so they are not separately identifiable to 1e-4 while the mapping itself is
PERSPECTIVE_BOUNDS = {"landscape": (0.08, 0.25, 1.0), "portrait": (0.16, 0.30, 1.0)}   # measured max (48 pairs): 0.026 / 0.170, 0.076 / 0.199


@pytest.fixture(scope="module")
def api(pkg):
    from vstab_amd import flow_pipeline, host_math

    class A:
        pass

    a = A()
    a.fp, a.hm = flow_pipeline, host_math
    return a


def _run(api, ctx, cam, w, h, mode, camera_lock=False, strength=0.7):
    import torch

    import bench

    frames = bench.synth_clip(cam.shape[0], 0, h, w, torch.device("cuda"), mats=cam)
    res = api.fp._stabilize_frames(api.hm._normalize_video_input(frames), "crop_and_pad", mode, camera_lock, strength, 0.5, 0.6,
                                   (127, 127, 127), 16.0, ctx=ctx, keep_on_device=True)
    return frames, res


@pytest.mark.parametrize("amp", [0.25, 1.0, 3.0])
@pytest.mark.parametrize("mode", ["translation", "similarity", "perspective"])
@pytest.mark.parametrize("size", [(960, 540), (1920, 1080), (540, 960)])
def test_known_camera_motion_is_recovered(api, ctx, size, mode, amp):
    import bench

    w, h = size
    cam = shake_path(N, w, h, mode, seed=3, amp=amp)
    _, res = _run(api, ctx, cam, w, h, mode)
    tr = res.meta["estimated_motion"]["per_transition"]
    assert [t["mode"] for t in tr] == [mode] * (N - 1)
    acc = bench.transition_accuracy([t["matrix"] for t in tr], cam, size, api.hm._working_estimation_size(w, h))
    shape = "portrait" if h > w else "landscape"
    centre, corner, lin = PERSPECTIVE_BOUNDS[shape] if mode == "perspective" else BOUNDS[(shape, mode)]
    assert acc["centre_px"]["max"] <= centre and acc["corner_px"]["max"] <= corner and acc["lin_2x2"]["max"] <= lin, acc
    # every pair moved: the bound is not met by reporting identity
    assert acc["true_motion_px"]["max"] > 0.5


@pytest.mark.parametrize("mode", ["translation", "similarity"])
@pytest.mark.parametrize("size", [(960, 540), (1920, 1080)])
def test_locked_camera_shows_the_static_texture(api, ctx, size, mode):
    """camera_lock=True, strength 1 (flow.py:356-371: the target path is 0, so frame i is moved back by its whole
    accumulated path) on a translation-only shake: every output frame must show frame 0's view of the texture, displaced
    by the crop_and_pad recentring offset the meta reports -- compared with that image sampled analytically, over the
    region no frame padded.  The error is bilinear interpolation of the texture plus the accumulated estimation error
    (a random walk of <= 0.05 px steps).  Measured: PSNR 51-55 dB, mean |error| 0.0023-0.0030 (of a 0..1 range)."""
    import torch

    import bench

    w, h = size
    cam = shake_path(N, w, h, "translation", seed=5, amp=1.0)
    frames, res = _run(api, ctx, cam, w, h, mode, camera_lock=True, strength=1.0)
    assert np.all(np.array(res.meta["estimated_motion"]["target_path"]) == 0.0) and res.meta["smooth"] == 0.85
    err = bench.static_texture_error(res, cam, frames, torch.device("cuda"))
    assert err["safe_fraction"] > 0.9
    assert err["psnr_db"] >= 48.0 and err["mean_abs"] <= 0.005 and err["max_abs"] <= 0.04, err
    # the input itself is NOT static: the same comparison on the unstabilised frames fails by a wide margin
    class Raw:
        pass

    raw = Raw()
    raw.frames, raw.masks, raw.meta = frames, torch.zeros_like(res.masks), {"framing": {"center_offset": [0.0, 0.0]}}
    assert bench.static_texture_error(raw, cam, frames, torch.device("cuda"))["psnr_db"] < 30.0


@pytest.mark.parametrize("interp", ["bilinear", "bicubic"])
@pytest.mark.parametrize("blur_samples", [0, 9])
def test_warp_reproduces_an_analytic_texture(ctx, interp, blur_samples):
    """warpPerspective's conventions without OpenCV or the oracle: warping frame(p) = T(p) by M must give T(M^-1 p) up to
    the interpolation error of the band-limited texture and the 1/32-px coordinate quantisation (1080p: measured max
    2-3e-3, mean 3-7e-4; a half-pixel convention error would be ~0.03, a transposed / inverted matrix far more).  M has
    sub-pixel translation, rotation, zoom and perspective.  With blur_samples the same matrix is given to every frame,
    so the S-sample motion blur (staged-window kernel) must reproduce the plain result."""
    import torch

    import bench

    w, h, n = 1920, 1080, 2
    dev = torch.device("cuda")
    src = bench.synth_clip(1, 0, h, w, dev, mats=np.eye(3)[None], seed=7).expand(n, h, w, 3).contiguous()
    m = np.array([[1.01 * np.cos(0.02), -1.01 * np.sin(0.02), 13.37], [1.01 * np.sin(0.02), 1.01 * np.cos(0.02), -9.61], [5e-6, -2.5e-6, 1.0]])
    want = bench.synth_clip(1, 0, h, w, dev, mats=m[None], seed=7)[0]
    mats = np.tile(m[None], (n, 1, 1))
    if blur_samples:
        got, mask = ctx.warp_blur_batch(src, mats, (w, h), 0.5, blur_samples, interp=interp, border=(0.5, 0.5, 0.5))
    else:
        got, mask, _ = ctx.warp_batch(src, mats.astype(np.float32), (w, h), interp=interp, border=(0.5, 0.5, 0.5))
    inside = (mask[0] == 0)
    inside = ~(torch.nn.functional.max_pool2d((~inside)[None, None].float(), 9, stride=1, padding=4)[0, 0] > 0)
    inside[:4] = inside[-4:] = False
    inside[:, :4] = inside[:, -4:] = False
    err = (got[0] - want).abs().amax(dim=-1)
    assert float(inside.float().mean()) > 0.9
    assert float(err[inside].max()) < 6e-3 and float(err[inside].mean()) < 1.5e-3, (float(err[inside].max()), float(err[inside].mean()))
