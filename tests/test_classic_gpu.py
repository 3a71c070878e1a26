"""GPU parity of the Classic estimator (sparse corners + pyramidal LK, SURVEY 8f N1 / BASELINE config C1)
against the oracle, through the C ABI.

Bar: corner coordinates / counts, tracked positions and status bytes are bit-exact (integer pixel arithmetic,
exact integer normal equations, same f32 operation order); the model fits follow tests/test_fit_gpu.py's
tolerances.  Parity against a real OpenCV is unpinned (oracle/vo_classic.c header).
"""

import json

import numpy as np
import pytest

from tests.test_dis_gpu import moving_clip
from tests.util import synth_frames

pytestmark = pytest.mark.gpu
BORDER = np.array([127, 127, 127], np.float32) / 255.0


def noisy_clip(n, h, w, seed):
    """Clip whose frames share little structure: exercises lost tracks, status 0 and window-out-of-range exits."""
    f = synth_frames(n, h, w, seed=seed)
    return (f[..., 1] * 255).astype(np.uint8)


@pytest.mark.parametrize("h,w,kind", [(480, 854, "moving"), (270, 480, "moving"), (120, 213, "moving"), (67, 91, "moving"),
                                      (48, 64, "noise"), (135, 240, "noise"), (40, 200, "flat")])
def test_gftt_matches_oracle(ctx, oracle, h, w, kind):
    import torch

    if kind == "moving":
        gray, _ = moving_clip(3, h, w, seed=h + w)
    elif kind == "noise":
        gray = noisy_clip(3, h, w, seed=h)
    else:
        gray = np.full((2, h, w), 93, np.uint8)
        gray[1, 10:30, 50:90] = 180   # one rectangle: a handful of corners, far fewer than 12
    corners, counts = ctx.gftt_batch(torch.from_numpy(gray).cuda())
    corners, counts = corners.cpu().numpy(), counts.cpu().numpy()
    for i in range(gray.shape[0]):
        ref = oracle.good_features(gray[i], **oracle.GFTT)
        assert counts[i] == ref.shape[0], (i, counts[i], ref.shape[0])
        assert np.array_equal(corners[i, : counts[i]], ref), i
    if kind == "flat":
        assert counts[0] == 0 and 0 < counts[1] < 12


@pytest.mark.parametrize("params", [dict(max_corners=50, quality=0.05, min_distance=3.0, block_size=7),
                                    dict(max_corners=1000, quality=0.001, min_distance=0.0, block_size=3),
                                    dict(max_corners=4096, quality=0.2, min_distance=15.0, block_size=31)])
def test_gftt_other_parameters(ctx, oracle, params):
    """The entry point is not specialised to the node's constants (no minimum distance, other block sizes)."""
    import torch

    gray, _ = moving_clip(2, 150, 260, seed=9)
    corners, counts = ctx.gftt_batch(torch.from_numpy(gray).cuda(), **params)
    corners, counts = corners.cpu().numpy(), counts.cpu().numpy()
    for i in range(2):
        ref = oracle.good_features(gray[i], params["max_corners"], params["quality"], params["min_distance"], params["block_size"])
        assert counts[i] == ref.shape[0] and np.array_equal(corners[i, : counts[i]], ref)


@pytest.mark.parametrize("h,w,kind", [(480, 854, "moving"), (270, 480, "moving"), (135, 240, "moving"), (67, 91, "moving"),
                                      (48, 64, "moving"), (135, 240, "noise"), (33, 40, "noise")])
def test_lk_matches_oracle(ctx, oracle, h, w, kind):
    import torch

    n = 3
    gray = moving_clip(n, h, w, seed=h * 3 + w)[0] if kind == "moving" else noisy_clip(n, h, w, seed=w)
    dgray = torch.from_numpy(gray).cuda()
    corners, counts = ctx.gftt_batch(dgray[:-1])
    pairs, nxt, status = ctx.lk_track_batch(dgray, corners, counts, want_raw=True)
    pairs, nxt, status, corners, counts = (t.cpu().numpy() for t in (pairs, nxt, status, corners, counts))
    assert ctx.lib.vstab_lk_levels(h, w, 31, 3) == oracle.lk_levels(h, w)
    for i in range(n - 1):
        k = int(counts[i])
        if k == 0:
            continue
        ref_next, ref_status = oracle.lk_track(gray[i], gray[i + 1], corners[i, :k], **oracle.LK)
        assert np.array_equal(status[i, :k], ref_status), (i, int((status[i, :k] != ref_status).sum()))
        assert np.array_equal(nxt[i, :k], ref_next), (i, np.abs(nxt[i, :k] - ref_next).max())
        ok = ref_status == 1
        assert np.array_equal(pairs[i, :k, :2], corners[i, :k])
        assert np.array_equal(pairs[i, :k, 2:][ok], ref_next[ok]) and np.all(np.isnan(pairs[i, :k, 2:][~ok]))
        assert np.all(status[i, k:] == 0)
    if kind == "moving" and h >= 135:
        assert status.sum() > 0.9 * counts.sum()   # the texture is trackable


def test_lk_points_near_and_outside_the_border(ctx, oracle):
    """Windows that hang over the image edge (reflected pixels, zero derivatives) and points whose window leaves
    the addressable range (status 0 at level 0): hand-placed points instead of detected corners."""
    import torch

    gray, _ = moving_clip(2, 120, 160, seed=5)
    pts = np.array([[0, 0], [159, 119], [3.25, 60.5], [156.75, 5.5], [80, 118.9], [-20.0, 50.0], [200.0, 40.0], [80.0, -47.0],
                    [80.5, 60.25], [12.0, 12.0], [150.0, 100.0], [40.0, 80.0]], np.float32)
    corners = torch.zeros((1, 16, 2), dtype=torch.float32, device="cuda")
    corners[0, : len(pts)] = torch.from_numpy(pts).cuda()
    counts = torch.tensor([len(pts)], dtype=torch.int32, device="cuda")
    _, nxt, status = ctx.lk_track_batch(torch.from_numpy(gray).cuda(), corners, counts, want_raw=True)
    ref_next, ref_status = oracle.lk_track(gray[0], gray[1], pts, **oracle.LK)
    assert np.array_equal(status.cpu().numpy()[0, : len(pts)], ref_status)
    assert np.array_equal(nxt.cpu().numpy()[0, : len(pts)], ref_next)
    assert 0 < ref_status.sum() < len(pts)


@pytest.mark.parametrize("mode", ["translation", "similarity", "perspective"])
def test_points_fit_matches_oracle(ctx, oracle, mode):
    """classic.py:98-160 on synthetic tracks: inliers under a known model + gross outliers + lost tracks;
    pairs with too few features (< 12) or too few tracked points (< 8) produce no candidate."""
    import torch

    from vstab_amd import native

    rng = np.random.default_rng(11)
    cap, cases = 400, []
    for count, lost, outl in [(400, 0.05, 0.2), (60, 0.3, 0.1), (11, 0.0, 0.0), (30, 0.8, 0.0), (12, 0.0, 0.0), (400, 0.0, 0.7)]:
        prev = rng.uniform(0, 1, (count, 2)) * [854, 480]
        th, s = rng.uniform(-0.02, 0.02), rng.uniform(0.98, 1.02)
        A = np.array([[s * np.cos(th), -s * np.sin(th)], [s * np.sin(th), s * np.cos(th)]])
        nxt = prev @ A.T + rng.uniform(-5, 5, 2) + rng.normal(0, 0.2, (count, 2))
        bad = rng.random(count) < outl
        nxt[bad] += rng.uniform(-60, 60, (int(bad.sum()), 2))
        status = (rng.random(count) >= lost).astype(np.uint8)
        cases.append((prev.astype(np.float32), nxt.astype(np.float32), status))
    table = np.full((len(cases), cap, 4), np.nan, np.float32)
    counts = np.zeros(len(cases), np.int32)
    for i, (prev, nxt, status) in enumerate(cases):
        k = prev.shape[0]
        counts[i] = k
        table[i, :k, :2] = prev
        table[i, :k, 2:] = np.where(status[:, None] == 1, nxt, np.nan)
    got = native.fit_table_to_dicts(ctx.points_fit_batch(torch.from_numpy(table).cuda(), torch.from_numpy(counts).cuda(), mode))
    for i, (prev, nxt, status) in enumerate(cases):
        ref, nv = oracle.fit_all_modes_points(prev, nxt, status, mode)
        assert set(got[i]) == set(ref), (i, set(got[i]), set(ref))
        if prev.shape[0] < 12 or status.sum() < 8:
            assert ref == {}
        for name, r in ref.items():
            g = got[i][name]
            assert g["valid_points"] == nv and g["total_points"] == prev.shape[0]
            assert g["accepted"] == r["accepted"] and g["confidence"] == r["confidence"], (i, name)
            if r["accepted"]:
                if name == "translation":
                    assert np.array_equal(g["matrix"], r["matrix"])
                elif name == "similarity":
                    assert np.allclose(g["matrix"], r["matrix"], rtol=0, atol=2e-6)
                else:
                    assert np.allclose(g["matrix"], r["matrix"], rtol=2e-5, atol=1e-7)


CLASSIC_META_KEYS = ["frames", "transform_mode_requested", "transform_mode_applied", "camera_lock", "strength", "strength_effective",
                     "smooth", "fps_requested", "fps_effective", "framing", "keep_fov_applied", "padding_color_rgb",
                     "stabilization_warp", "estimated_motion", "padding_fraction_mean", "padding_fraction_max", "motion_meta"]


@pytest.mark.parametrize("size,n,mode,framing", [((854, 480), 64, "translation", "crop_and_pad"),      # BASELINE config C1
                                                 ((640, 480), 8, "similarity", "expand"),
                                                 ((1280, 720), 5, "perspective", "crop_and_pad"),     # estimated at 960x540
                                                 ((480, 270), 6, "similarity", "crop")])
def test_classic_pipeline_against_oracle(pkg, ctx, oracle, size, n, mode, framing):
    """The Classic node end to end (classic.py:163-570): meta key list / lengths / types (no flow_backend, no
    residual, source "estimated_classic"), every transition against the oracle's GFTT -> LK -> fit chain, known
    motion recovered, warp against the oracle with the node's matrices, and the bit-identical Motion Apply replay."""
    from vstab_amd import apply_pipeline as ap
    from vstab_amd import flow_pipeline as fp
    from vstab_amd import host_math as hm

    w, h = size
    gray_full, params = moving_clip(n, h, w, seed=w + n)
    frames = np.repeat(gray_full[..., None].astype(np.float32) / 255.0, 3, axis=-1)
    frames[..., 2] *= 0.8
    frames = np.ascontiguousarray(frames)
    res = fp._stabilize_frames(hm._normalize_video_input(frames), framing, mode, False, 0.7, 0.5, 0.6, (127, 127, 127), 16.0,
                               estimator="classic")
    meta = res.meta
    assert list(meta.keys()) == CLASSIC_META_KEYS
    em = meta["estimated_motion"]
    pdim = {"translation": 2, "similarity": 4, "perspective": 8}[mode]
    assert meta["frames"] == n and len(em["per_transition"]) == n - 1
    assert all(list(t.keys()) == ["index", "mode", "confidence", "matrix"] for t in em["per_transition"])
    assert [t["index"] for t in em["per_transition"]] == list(range(n - 1))
    assert np.array(em["path"]).shape == (n, pdim) and np.array(em["target_path_effective"]).shape == (n, pdim)
    assert meta["motion_meta"]["source"] == "estimated_classic" and meta["motion_meta"]["frame_count"] == n
    assert len(meta["stabilization_warp"]["per_frame"]) == n
    json.dumps(meta)
    # --- estimator vs oracle (sticky mode walk on the oracle's candidates)
    work = hm._working_estimation_size(w, h)
    g = oracle.gray_for_estimation(frames, work)
    recs = []
    for i in range(n - 1):
        feats = oracle.good_features(g[i], **oracle.GFTT)
        if feats.shape[0] < 12:
            recs.append({})
            continue
        nxt, status = oracle.lk_track(g[i], g[i + 1], feats, **oracle.LK)
        recs.append(oracle.fit_all_modes_points(feats, nxt, status, mode)[0])
    mats, modes, confs, _, active = fp.select_transitions(recs, mode)
    assert meta["transform_mode_applied"] == active
    for i, t in enumerate(em["per_transition"]):
        assert t["mode"] == modes[i] and t["confidence"] == confs[i], i
        full = hm._rescale_transform_to_full(mats[i], (w, h), work) if work else mats[i]
        tol = dict(rtol=2e-5, atol=1e-6) if mode == "perspective" else dict(rtol=0, atol=2e-5 if work else 2e-6)
        assert np.allclose(np.array(t["matrix"], np.float32), full, **tol), i
    # --- known motion (independent of the oracle)
    def to_texture(pr):
        tx, ty, th, sc = pr
        c, sn = np.cos(th) / sc, np.sin(th) / sc
        lin = np.array([[c, sn, 0], [-sn, c, 0], [0, 0, 1.0]])
        return np.array([[1, 0, w / 2], [0, 1, h / 2], [0, 0, 1.0]]) @ lin @ np.array([[1, 0, -w / 2 - tx], [0, 1, -h / 2 - ty], [0, 0, 1.0]])

    for i, t in enumerate(em["per_transition"]):
        expect = np.linalg.inv(to_texture(params[i + 1])) @ to_texture(params[i])
        got = np.array(t["matrix"])
        assert np.abs(got[:2, :2] - expect[:2, :2]).max() < (2e-2 if mode == "translation" else 3e-3)
        assert np.abs(got[:2, 2] - expect[:2, 2]).max() < (2.5 if mode == "translation" else 0.8)
    # --- warp with the node's matrices vs oracle; KA7 replay
    fm = np.array([e["applied_matrix"] for e in meta["stabilization_warp"]["per_frame"]], np.float32)
    out_size = tuple(meta["stabilization_warp"]["output_size"])
    ref, ref_mask, cnt = oracle.warp_clip(frames, fm, out_size, border=BORDER)
    assert np.array_equal(res.frames, ref) and np.array_equal(res.masks[..., 0], ref_mask)
    if framing != "crop":
        replay = ap.apply_motion(hm._normalize_video_input(frames), meta, (127, 127, 127), framing_mode="crop_and_pad")
        assert np.array_equal(replay.frames, res.frames) and np.array_equal(replay.masks, res.masks)
    else:
        assert float(res.masks.max()) == 0.0


def test_classic_node_small_paths(pkg, ctx):
    """Single frame, featureless clip (identity transitions reported as translation / confidence 0, classic.py:84-86)
    and the crop bypass -- with the Classic flavour of each meta."""
    import torch

    from vstab_amd import nodes

    one = synth_frames(1, 60, 80)
    out = nodes.VideoStabilizerClassic.execute(torch.from_numpy(one), 16.0, "crop_and_pad", "similarity", False, 0.7, 0.5, 0.6, "#7F7F7F")
    assert np.array_equal(out[0].numpy(), one)
    assert list(out[2].keys()) == ["frames", "note", "transform_mode", "framing_mode", "stabilization_warp", "fps_requested",
                                   "fps_effective", "motion_meta"]
    assert out[2]["motion_meta"]["source"] == "estimated_classic"
    flat = np.full((4, 64, 96, 3), 0.4, np.float32)
    out = nodes.VideoStabilizerClassic.execute(torch.from_numpy(flat), 16.0, "crop_and_pad", "perspective", False, 0.7, 0.5, 0.6, "#7F7F7F")
    per = out[2]["estimated_motion"]["per_transition"]
    assert all(t["mode"] == "translation" and t["confidence"] == 0.0 and np.array_equal(np.array(t["matrix"]), np.eye(3)) for t in per)
    assert out[2]["transform_mode_applied"] == "translation" and np.array_equal(out[0].numpy(), flat)
    clip = synth_frames(4, 72, 128)
    out = nodes.VideoStabilizerClassic.execute(torch.from_numpy(clip), 16.0, "crop", "similarity", False, 0.7, 0.5, 1.0, "#7F7F7F")
    assert out[2]["note"].startswith("keep_fov~=1.0") and "flow_backend" not in out[2] and np.array_equal(out[0].numpy(), clip)
    assert out[2]["motion_meta"]["source"] == "estimated_classic"
