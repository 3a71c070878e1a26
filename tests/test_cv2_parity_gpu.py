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


@pytest.mark.gpu
def test_bench_cv2_leg_computes_its_parity_object_under_the_standin(ctx, pkg, oracle):
    """The code of bench.cv2_leg that will run the day a cv2 imports -- the baseline through the tier, the HIP run on the same
    frames, every entry of `cv2_parity` -- exercised now with the oracle-backed stand-in as the `cv2` (allow_standin: never
    done by the bench itself).  The stand-in answers with the oracle's bits and the HIP path equals the oracle, so every
    difference is zero except under the OTHER sub-pixel convention, and `q5` is reported as the one that matched."""
    import sys

    import torch

    import bench
    from tests.golden import cv2_standin

    saved = sys.modules.get("cv2")
    cv2_standin.install()
    try:
        frames = bench.synth_clip(6, 0, 1080, 1920, ctx.device)
        leg = bench.cv2_leg(ctx, torch, frames, 4, 6, allow_standin=True)
    finally:
        if saved is None:
            sys.modules.pop("cv2", None)
        else:
            sys.modules["cv2"] = saved
    assert leg["probe"]["standin"] is True and leg["baseline"]["kind"] == "opencv" and leg["baseline"]["value"] > 0
    par = leg["parity"]
    assert par["frames"] == 6 and par["gray_u8_differing"] == 0 and par["flow_epe_px_on_cv2_gray"]["max"] == 0.0
    assert par["transition_matrices_max_abs"] == 0.0 and par["final_matrices_max_abs"] == 0.0
    assert par["warp_pixels_q5"]["max"] == 0.0 and par["mask_pixels_differing"] == 0 and par["end_to_end_pixels"]["max"] == 0.0
    assert par["warp_pixels_exact"]["max"] > 0.0 and par["subpix_matched"] == "q5" and par["within_1e-3"] is True
