"""HIP path vs a REAL OpenCV on the GPU box (skips while no cv2 imports there): bench.cv2_leg on a short 1080p clip --
the same object the bench line carries as `cv2_parity` -- held to north_star's bound (pixels within 1e-3 under the
sub-pixel convention that OpenCV build uses, sampled flow within 1e-3 px)."""

import pytest


@pytest.mark.gpu
def test_hip_path_against_a_real_opencv(ctx, pkg):
    import torch

    import bench
    from oracle import cv2_tier

    if not cv2_tier.available(allow_standin=False):
        pytest.skip("no real cv2 importable on this box: " + str(cv2_tier.probe()))
    frames = bench.synth_clip(12, 0, 1080, 1920, ctx.device)
    leg = bench.cv2_leg(ctx, torch, frames, 8, 12)
    print(leg)
    par = leg["parity"]
    assert leg["baseline"]["kind"] == "opencv"
    assert par["gray_u8_max_abs"] <= 1
    assert par["flow_epe_px_on_cv2_gray"]["at_stride8_max"] <= 1e-3
    assert par["within_1e-3"] and par["mask_pixels_differing"] == 0


@pytest.mark.gpu
def test_bench_leg_reports_absent_without_cv2(ctx, pkg):
    import torch

    import bench
    from oracle import cv2_tier

    if cv2_tier.available(allow_standin=False):
        pytest.skip("a real cv2 is importable")
    leg = bench.cv2_leg(ctx, torch, torch.zeros((2, 64, 64, 3), device=ctx.device), 2, 2)
    assert leg == {"probe": {"cv2": "absent"}}
