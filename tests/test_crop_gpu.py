"""GPU tests for framing_mode="crop": vstab_crop_analysis vs the oracle (bit-exact integers) and the
reference-pinned properties of crop mode (KA8, scripts/check_crop_aspect_ratio.py:82-120,173-233)."""

import numpy as np
import pytest

from tests.util import test_matrices as make_matrices

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kind", ["similarity", "perspective", "translation", "far"])
@pytest.mark.parametrize("size", [(73, 45), (212, 120)])
def test_crop_analysis_matches_oracle(ctx, oracle, kind, size):
    w, h = size
    mats = make_matrices(6 if kind != "far" else 2, w, h, kind).astype(np.float32)
    ref_bbox, ref_common = oracle.crop_analysis(mats, (w, h), (w, h))
    bbox, common = ctx.crop_analysis(mats, (w, h), (w, h))
    assert np.array_equal(bbox, ref_bbox) and np.array_equal(common, ref_common)


def test_largest_rectangle_search(pkg):
    from vstab_amd import crop_solver as cs

    mask = np.zeros((45, 73), np.uint8)
    mask[5:40, 6:70] = 1
    x0, y0, cw, ch = cs._largest_aspect_ratio_rectangle(mask, 73, 45)
    assert abs(cw / ch - 73 / 45) < 1e-9 and mask[int(y0):int(y0 + ch), int(x0):int(np.ceil(x0 + cw))].all()
    assert ch == 35.0 or int(np.ceil(73 / 45 * (ch + 1))) > 64  # cannot grow further
    assert cs._largest_aspect_ratio_rectangle(np.zeros((10, 10), np.uint8), 10, 10) is None


@pytest.mark.parametrize("mode", ["translation", "similarity"])
@pytest.mark.parametrize("keep_fov", [0.0, 0.6])
def test_flow_crop_mode_properties(pkg, ctx, mode, keep_fov):
    """KA8: crop mode has a zero padding mask, the crop keeps the frame aspect (1e-6) and a uniform scale."""
    from tests.test_dis_gpu import moving_clip
    from vstab_amd import flow_pipeline as fp
    from vstab_amd import host_math as hm

    w, h, n = 480, 270, 8
    gray, _ = moving_clip(n, h, w, seed=11)
    frames = np.ascontiguousarray(np.repeat(gray[..., None].astype(np.float32) / 255.0, 3, axis=-1))
    res = fp._stabilize_frames(hm._normalize_video_input(frames), "crop", mode, False, 0.7, 0.5, keep_fov, (127, 127, 127), 16.0)
    assert res.frames.shape == (n, h, w, 3)
    assert float(res.masks.max()) == 0.0 and res.meta["padding_fraction_max"] == 0.0
    fr = res.meta["framing"]
    assert fr["mode"] == "crop" and fr["keep_fov_status"] in ("met", "clamped", "disabled", "failed")
    cw, ch = fr["crop_size"]
    assert abs(cw / ch - w / h) < 1e-6
    assert fr["keep_fov_effective"] == 1.0 and fr["actual_content_ratio"] == 1.0
    if keep_fov > 1e-6:
        assert fr["keep_fov_requested"] == keep_fov and res.meta["keep_fov_applied"] is True
    for entry in res.meta["stabilization_warp"]["per_frame"]:
        m = np.array(entry["applied_matrix"])
        if mode == "translation":
            assert abs(m[0, 0] - m[1, 1]) < 1e-6 and abs(m[0, 1]) < 1e-6 and abs(m[1, 0]) < 1e-6  # uniform scale, no shear
    assert list(res.meta["framing"].keys())[:4] == ["mode", "input_size", "padding_color_rgb", "min_content_ratio"]
    import json

    json.dumps(res.meta)
