"""CPU: the oracle (the checker of every GPU parity test) on the edge inputs of the GPU stress tests -- degenerate warp
maps, DIS on flat / noise / jump content, out-of-range gray samples.  Checks invariants that need no GPU, and serves as
the workload of the sanitizer run (tools/oracle_sanitize.sh: AddressSanitizer + UBSan build of oracle/*.c)."""

import numpy as np
import pytest

from tests.util import synth_frames, test_matrices as make_matrices

BORDER = (np.array([127, 127, 127], np.float32) / 255.0)


@pytest.mark.parametrize("interp", ["bilinear", "bicubic"])
@pytest.mark.parametrize("kind", ["horizon", "flip", "minify", "magnify", "quarter_turn", "far"])
def test_warp_oracle_on_stress_maps(oracle, kind, interp):
    frames = synth_frames(3, 45, 73, seed=11)
    mats = make_matrices(3, 73, 45, kind).astype(np.float32)
    out, mask, cnt = oracle.warp_clip(frames, mats, (80, 51), interp=interp, border=BORDER)
    assert out.shape == (3, 51, 80, 3) and np.isfinite(out).all()
    assert set(np.unique(mask)) <= {0.0, 1.0}
    assert np.array_equal(cnt, (mask > 0.5).reshape(3, -1).sum(axis=1).astype(cnt.dtype))
    # a fully padded pixel carries the border colour exactly
    far_out = mask > 0.5
    if kind in ("far", "minify") and interp == "bilinear":
        assert far_out.mean() > 0.9
    if kind == "magnify":
        assert not far_out.any() and out.min() >= frames.min() - 0.3 and out.max() <= frames.max() + 0.3
    blur, bmask = oracle.warp_blur_clip(frames, make_matrices(3, 73, 45, kind), (73, 45), 0.7, 5, interp=interp, border=BORDER)
    assert np.isfinite(blur).all() and bmask.min() >= 0.0 and bmask.max() <= 1.0


@pytest.mark.parametrize("name", ["flat", "noise", "bars", "jump", "still", "saturated", "half_noise"])
def test_dis_oracle_on_content_edge_cases(oracle, name):
    from tests.test_dis_gpu import _content_clips

    gray = np.ascontiguousarray(_content_clips(96, 128)[name])
    flow = oracle.dis_flow_clip(gray)
    assert flow.shape == (2, 96, 128, 2) and np.isfinite(flow).all()
    if name in ("flat", "still"):
        assert np.abs(flow[-1]).max() < 1e-3
    fits = oracle.fit_all_modes(flow[0], 8, "perspective")[0]
    assert "translation" in fits and np.isfinite(fits["translation"]["matrix"]).all()


def test_gray_oracle_on_out_of_range_samples(oracle):
    rng = np.random.default_rng(5)
    frames = rng.uniform(-0.6, 1.8, (2, 40, 1000, 3)).astype(np.float32)
    frames[0, 1, 2::7, 0] = np.inf
    frames[0, 2, 3::5, 2] = -np.inf
    frames[1, 3, ::9, 1] = 1.0e30
    g = oracle.gray_for_estimation(frames, (500, 20))
    assert g.dtype == np.uint8 and g.shape == (2, 20, 500)
    plain = oracle.gray_for_estimation(np.clip(np.nan_to_num(frames, posinf=2.0, neginf=-1.0), -1.0, 2.0), None)
    assert plain.min() == 0 and plain.max() == 255
    assert np.array_equal(oracle.frame_max(frames), frames.reshape(2, -1).max(axis=1))
