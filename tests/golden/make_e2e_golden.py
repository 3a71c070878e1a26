#!/usr/bin/env python3
"""End-to-end golden fixtures from the REFERENCE's own pipelines, run in the build container.

    python tests/golden/make_e2e_golden.py          (needs /root/reference; writes tests/golden/e2e/*.npz + *.json)

What runs: the reference's `_stabilize_frames` (nodes/video_stabilizer_flow.py:213-640) and `apply_motion`
(nodes/motion_apply.py:297-429), imported from /root/reference, unmodified, under the `cv2` stand-in of
tests/golden/cv2_standin.py whose primitives are answered by the CPU oracle.  So every line of the reference's host
logic executes -- the per-pair loop with the sticky active_mode (flow.py:324-352), working-size rescale, parameter
deltas, path / target (flow.py:356-371), the three framing branches incl. the keep_fov crop solver
(stabilizer_utils.py:518-837), the warp loop's mask statistics, meta assembly (flow.py:596-640), Motion Apply's
resolve / crop / expand / blur accumulation and its progress ticks.

What the fixtures are: DATA -- uint8 input frames, the float32 frames / masks the reference produced (whole for the
tiny clips, a fixed sub-grid for the larger ones), the meta dict as JSON, the ProgressBar call sequence.  No reference
code travels.  What they pin: the build's host chain and kernels' agreement with "reference control flow over oracle
primitives".  What they do not pin: the OpenCV primitives themselves (still the oracle's restatement).
"""

from __future__ import annotations

import json
import sys
import types
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
ROOT = HERE.parents[1]
REF = Path("/root/reference")
OUT = HERE / "e2e"
sys.path.insert(0, str(ROOT))

from tests.golden import cv2_standin  # noqa: E402

PROGRESS: list = []


def _install_comfy_stubs() -> None:
    class ProgressBar:
        def __init__(self, total):
            PROGRESS.append(("init", int(total)))

        def update_absolute(self, value, total=None):
            PROGRESS.append((int(value), None if total is None else int(total)))

    comfy = types.ModuleType("comfy")
    comfy_utils = types.ModuleType("comfy.utils")
    comfy_utils.ProgressBar = ProgressBar
    comfy.utils = comfy_utils
    sys.modules["comfy"] = comfy
    sys.modules["comfy.utils"] = comfy_utils
    api = types.ModuleType("comfy_api")
    latest = types.ModuleType("comfy_api.latest")
    latest.ComfyExtension = type("ComfyExtension", (), {})
    latest.io = types.SimpleNamespace(Custom=lambda k: types.SimpleNamespace(Input=lambda *a, **k: None, Output=lambda *a, **k: None),
                                      ComfyNode=object)
    api.latest = latest
    sys.modules["comfy_api"] = api
    sys.modules["comfy_api.latest"] = latest


# ---- synthetic clips (own construction; stored as uint8 so that no libm / SIMD difference can change an input) -----
def _texture(xx, yy, seed):
    r = np.random.default_rng(seed)
    v = np.zeros_like(xx)
    for _ in range(40):
        fx, fy = r.uniform(-0.3, 0.3, 2)
        v += r.uniform(0.3, 1.0) * np.sin(fx * xx + fy * yy + r.uniform(0, 6.28))
    return (v - v.min()) / (v.max() - v.min())


def moving_rgb_clip(n, h, w, seed, step=(2.2, 1.4), rot=0.004, zoom=0.003, persp=0.0):
    """RGB texture under a random-walk camera (similarity, optionally a little perspective): uint8 [n,h,w,3]."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    tx = ty = th = 0.0
    s = 1.0
    px = py = 0.0
    out = np.empty((n, h, w, 3), np.uint8)
    for i in range(n):
        if i:
            tx += rng.uniform(-step[0], step[0])
            ty += rng.uniform(-step[1], step[1])
            th += rng.uniform(-rot, rot)
            s *= rng.uniform(1 - zoom, 1 + zoom)
            px += rng.uniform(-persp, persp)
            py += rng.uniform(-persp, persp)
        c, sn = np.cos(th) / s, np.sin(th) / s
        den = 1.0 + px * (xx - w / 2) + py * (yy - h / 2)
        X = (c * (xx - w / 2 - tx) + sn * (yy - h / 2 - ty)) / den + w / 2
        Y = (-sn * (xx - w / 2 - tx) + c * (yy - h / 2 - ty)) / den + h / 2
        for ch in range(3):
            out[i, ..., ch] = np.clip(_texture(X, Y, 100 * seed + ch) * 255.0, 0, 255).astype(np.uint8)
    return out


def as_float(u8):
    return (u8.astype(np.float32) / np.float32(255.0)).astype(np.float32)


def jsonable(x):
    if isinstance(x, np.ndarray):
        return x.tolist()
    if isinstance(x, (np.floating,)):
        return float(x)
    if isinstance(x, (np.integer,)):
        return int(x)
    if isinstance(x, (np.bool_,)):
        return bool(x)
    if isinstance(x, dict):
        return {str(k): jsonable(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [jsonable(v) for v in x]
    return x


def main() -> None:
    cv2_standin.install()
    _install_comfy_stubs()
    sys.path.insert(0, str(REF))
    from nodes import motion_apply as ma  # noqa: E402
    from nodes import motion_meta as mm  # noqa: E402
    from nodes import stabilizer_utils as su  # noqa: E402
    from nodes import video_stabilizer_classic as vc  # noqa: E402
    from nodes import video_stabilizer_flow as vf  # noqa: E402

    assert vf.cv2 is cv2_standin and su.cv2 is cv2_standin and ma.cv2 is cv2_standin and vc.cv2 is cv2_standin
    OUT.mkdir(exist_ok=True)
    index = {"flow": [], "apply": [], "note": "reference control flow over oracle primitives; see make_e2e_golden.py"}

    from nodes import shake_noise as sn  # noqa: E402

    tiny = moving_rgb_clip(8, 45, 73, seed=1, step=(0.9, 0.5), rot=0.01, zoom=0.002)
    mid = moving_rgb_clip(8, 64, 96, seed=2, step=(1.2, 0.8), rot=0.006, zoom=0.003)
    persp = moving_rgb_clip(5, 270, 480, seed=3, step=(3.0, 2.0), rot=0.004, zoom=0.003, persp=4e-6)
    # 1024x576 clip stored at half size: every pixel replicated 2x2 (exact, so the fixture holds 1/4 of the bytes)
    big_half = moving_rgb_clip(3, 288, 512, seed=4, step=(2.5, 1.5), rot=0.003, zoom=0.002)
    # a clip with two damaged frames: the frame is cut into tiles and every tile shows the scene under its own large
    # shift, so no global model explains more than a few percent of the flow samples.  Mild damage (30x30 tiles, +-14 px)
    # makes the homography fail its 0.15 inlier gate while the similarity still passes 0.1; strong damage (18x20 tiles,
    # +-24 px) rejects both (flow.py:173,187) -- with the sticky active_mode (flow.py:338-339) a perspective request walks
    # perspective -> similarity -> translation, a similarity request similarity -> translation.
    broken = moving_rgb_clip(7, 270, 480, seed=5, step=(1.5, 1.0))
    rng = np.random.default_rng(99)
    for frame, (th, tw, amp) in ((2, (30, 30, 14)), (4, (18, 20, 24))):
        src = broken[frame].copy()
        for by in range(0, 270, th):
            for bx in range(0, 480, tw):
                dy, dx = rng.integers(-amp, amp + 1, 2)
                broken[frame, by:by + th, bx:bx + tw] = np.roll(src, (int(dy), int(dx)), (0, 1))[by:by + th, bx:bx + tw]
    shake = sn.generate_shake_motion_meta(recipe=sn.STYLES["handheld"], frame_count=10, width=160, height=90, fps=16.0, amount=1.5,
                                          speed=1.0, seed=3, node="shake_generator", style="handheld")
    shake_clip = moving_rgb_clip(10, 90, 160, seed=6, step=(1.0, 0.7))
    clips = {"tiny": tiny, "mid": mid, "persp": persp, "big_half": big_half, "broken": broken, "shake": shake_clip}

    def clip_frames(name):
        u8 = clips[name]
        if name == "big_half":
            u8 = np.repeat(np.repeat(u8, 2, axis=1), 2, axis=2)
        return as_float(u8)

    flow_cases = [
        # the three Flow scenarios of scripts/compare_refactor_behavior.py:380-393 (8 x 73x45 @ 24 fps)
        ("tiny_crop_and_pad_similarity", "tiny", 1, ("crop_and_pad", "similarity", False, 0.7, 0.5, 0.6, (127, 127, 127), 24.0)),
        ("tiny_expand_translation", "tiny", 1, ("expand", "translation", False, 0.7, 0.5, 0.6, (127, 127, 127), 24.0)),
        ("tiny_crop_keep_fov_bypass", "tiny", 1, ("crop", "translation", False, 0.7, 0.5, 1.0, (127, 127, 127), 24.0)),
        # crop framing through the keep_fov solver (check_crop_aspect_ratio.py:82-120 uses keep_fov 0.6 and 0.0)
        ("mid_crop_keep_fov_0.6_similarity", "mid", 1, ("crop", "similarity", False, 0.7, 0.5, 0.6, (127, 127, 127), 16.0)),
        ("mid_crop_keep_fov_0_translation", "mid", 1, ("crop", "translation", False, 1.0, 1.0, 0.0, (10, 20, 30), 16.0)),
        ("mid_expand_camera_lock", "mid", 1, ("expand", "similarity", True, 0.7, 0.2, 0.6, (255, 0, 16), 30.0)),
        # perspective estimator (BASELINE config C3's Flow half at 480x270, no working-size downscale)
        ("persp_crop_and_pad_perspective", "persp", 4, ("crop_and_pad", "perspective", False, 0.7, 0.5, 0.6, (127, 127, 127), 16.0)),
        # working-size path: 1024x576 is estimated at 960x540 (general INTER_AREA ratio) and rescaled (flow.py:340-341)
        ("big_crop_and_pad_similarity", "big_half", 8, ("crop_and_pad", "similarity", False, 0.7, 0.5, 0.6, (127, 127, 127), 16.0)),
        # sticky mode downgrade
        ("broken_sticky_similarity", "broken", 4, ("crop_and_pad", "similarity", False, 0.7, 0.5, 0.6, (127, 127, 127), 16.0)),
        ("broken_sticky_perspective", "broken", 4, ("expand", "perspective", False, 1.0, 0.0, 0.6, (0, 0, 0), 16.0)),
        # the Classic node's pipeline (nodes/video_stabilizer_classic.py:163-568: GFTT + pyramidal LK per
        # pair, its own fallback order and meta keys) -- "estimator": "classic"
        ("classic_mid_crop_and_pad_similarity", "mid", 1, ("crop_and_pad", "similarity", False, 0.7, 0.5, 0.6, (127, 127, 127), 16.0), "classic"),
        ("classic_persp_expand_perspective", "persp", 4, ("expand", "perspective", False, 0.9, 0.3, 0.6, (16, 32, 64), 24.0), "classic"),
        ("classic_shake_crop_translation", "shake", 2, ("crop", "translation", False, 0.7, 0.5, 0.6, (127, 127, 127), 16.0), "classic"),
        ("classic_tiny_too_few_corners", "tiny", 1, ("crop_and_pad", "similarity", False, 0.7, 0.5, 0.6, (127, 127, 127), 16.0), "classic"),
        # the Flow node on its fallback estimator (flow.py:90-130: DIS cannot be created, cv2.optflow is
        # missing -> phase correlation, every pair reported as "translation") -- "estimator": "flow_phase_correlate"
        ("phase_shake_expand_similarity", "shake", 2, ("expand", "similarity", False, 0.7, 0.5, 0.6, (127, 127, 127), 16.0), "flow_phase_correlate"),
        ("phase_mid_crop_and_pad_translation", "mid", 1, ("crop_and_pad", "translation", False, 1.0, 0.2, 0.6, (127, 127, 127), 30.0), "flow_phase_correlate"),
    ]
    results = {}
    for case in flow_cases:
        name, clip, stride, args = case[:4]
        estimator = case[4] if len(case) > 4 else "flow"
        PROGRESS.clear()
        cv2_standin.CALLS.clear()
        frames = clip_frames(clip)
        ctx = su._normalize_video_input(frames)
        if estimator == "classic":
            res = vc._stabilize_frames(ctx, *args)
        elif estimator == "flow_phase_correlate":
            cv2_standin.DIS_DISABLED = "disabled by VSTAB_FLOW_BACKEND"   # the text this build's fallback reason quotes
            try:
                res = vf._stabilize_frames(ctx, *args)
            finally:
                cv2_standin.DIS_DISABLED = None
        else:
            res = vf._stabilize_frames(ctx, *args)
        out_frames = np.asarray(res.frames, dtype=np.float32)
        out_masks = np.asarray(res.masks, dtype=np.float32)
        meta = jsonable(res.meta)
        results[name] = (frames, res.meta)
        np.savez_compressed(OUT / f"flow_{name}.npz", frames=out_frames[:, ::stride, ::stride], masks=out_masks[:, ::stride, ::stride])
        (OUT / f"flow_{name}.json").write_text(json.dumps({
            "clip": clip, "args": jsonable(list(args)), "estimator": estimator, "stride": stride, "out_shape": list(out_frames.shape),
            "meta": meta, "progress": jsonable(PROGRESS), "cv2_calls": dict(cv2_standin.CALLS)}))
        index["flow"].append(name)
        modes = [t["mode"] for t in meta.get("estimated_motion", {}).get("per_transition", [])]
        print(f"flow  {name}: out {out_frames.shape} modes {sorted(set(modes))} applied {meta.get('transform_mode_applied')}"
              f" calls {dict(cv2_standin.CALLS)}")

    # ---- Motion Apply -------------------------------------------------------------------------------------------
    apply_cases = [
        # BASELINE config C3's second half in small: the Flow(perspective) meta replayed on the ORIGINAL frames,
        # bicubic, blur 0.5, High (17 samples; video_stabilizer_motion_apply.py:21-26)
        ("persp_replay_bicubic_blur17", "persp", 4, results["persp_crop_and_pad_perspective"][1], (127, 127, 127),
         dict(framing_mode="crop_and_pad", interpolation="bicubic", motion_blur=0.5, motion_blur_samples=17)),
        # BASELINE config C5 in small: expand framing + blur Ultra (33)
        ("shake_expand_blur33", "shake", 2, {"motion_meta": shake}, (127, 127, 127),
         dict(framing_mode="expand", interpolation="bilinear", motion_blur=0.5, motion_blur_samples=33)),
        ("shake_crop_plain", "shake", 2, {"motion_meta": shake}, (1, 2, 3),
         dict(framing_mode="crop", interpolation="bilinear", motion_blur=0.0, motion_blur_samples=9)),
        ("shake_crop_blur5_bicubic", "shake", 2, {"motion_meta": shake}, (1, 2, 3),
         dict(framing_mode="crop", interpolation="bicubic", motion_blur=1.0, motion_blur_samples=5)),
        ("shake_pad_alias_blur_clamped", "shake", 2, {"motion_meta": shake}, (200, 100, 50),
         dict(framing_mode="pad", interpolation="bilinear", motion_blur=3.0, motion_blur_samples=99)),
        # KA7 (check_crop_aspect_ratio.py:123-161): replay of the stabilizer's own meta
        ("tiny_replay_expand", "tiny", 1, results["tiny_expand_translation"][1], (127, 127, 127),
         dict(framing_mode="crop_and_pad", interpolation="bilinear", motion_blur=0.0, motion_blur_samples=9)),
        # legacy block only -> inverse selected by the connected frames' size (motion_apply.py:59-66)
        ("mid_legacy_inverse", None, 1, {"stabilization_warp": results["mid_expand_camera_lock"][1]["stabilization_warp"]}, (127, 127, 127),
         dict(framing_mode="crop_and_pad", interpolation="bilinear", motion_blur=0.0, motion_blur_samples=9)),
    ]
    for name, clip, stride, meta_in, rgb, kw in apply_cases:
        PROGRESS.clear()
        cv2_standin.CALLS.clear()
        if clip is None:   # frames of the expanded size: a synthetic clip of exactly that geometry
            ow, oh = meta_in["stabilization_warp"]["output_size"]
            clip = "legacy"
            clips[clip] = moving_rgb_clip(len(meta_in["stabilization_warp"]["per_frame"]), oh, ow, seed=7, step=(1.0, 1.0))
        frames = clip_frames(clip)
        ctx = su._normalize_video_input(frames)
        ticks = []
        res = ma.apply_motion(ctx, meta_in, rgb, progress_callback=lambda: ticks.append(1), **kw)
        out_frames = np.asarray(res.frames, dtype=np.float32)
        out_masks = np.asarray(res.masks, dtype=np.float32)
        np.savez_compressed(OUT / f"apply_{name}.npz", frames=out_frames[:, ::stride, ::stride], masks=out_masks[:, ::stride, ::stride])
        (OUT / f"apply_{name}.json").write_text(json.dumps({
            "clip": clip, "padding_rgb": list(rgb), "kwargs": kw, "stride": stride, "out_shape": list(out_frames.shape), "meta_in": jsonable(meta_in),
            "meta": jsonable(res.meta), "ticks": len(ticks), "cv2_calls": dict(cv2_standin.CALLS)}))
        index["apply"].append(name)
        print(f"apply {name}: out {out_frames.shape} ticks {len(ticks)} apply_meta {res.meta['motion_apply']}"
              f" fallback {res.meta.get('framing_fallback')}")

    # KA5: no common region -> framing_fallback (check_motion_meta.py:396-415), on literal inputs
    two = as_float(tiny[:2])
    blk = {"motion_meta": mm.build_motion_meta_v2(source="manual", frame_count=2, fps=16.0, input_size=(73, 45), output_size=(73, 45),
                                                  matrices=[np.eye(3), np.array([[1, 0, 150.0], [0, 1, 0], [0, 0, 1]])])}
    res = ma.apply_motion(su._normalize_video_input(two), blk, (127, 127, 127), framing_mode="crop")
    np.savez_compressed(OUT / "apply_tiny_crop_fallback.npz", frames=np.asarray(res.frames, np.float32), masks=np.asarray(res.masks, np.float32))
    (OUT / "apply_tiny_crop_fallback.json").write_text(json.dumps({
        "clip": "tiny", "clip_frames": 2, "padding_rgb": [127, 127, 127], "kwargs": dict(framing_mode="crop", interpolation="bilinear", motion_blur=0.0, motion_blur_samples=9),
        "stride": 1, "out_shape": list(np.asarray(res.frames).shape), "meta_in": jsonable(blk), "meta": jsonable(res.meta), "ticks": None,
        "cv2_calls": {}}))
    index["apply"].append("tiny_crop_fallback")
    np.savez_compressed(OUT / "clips.npz", **clips)
    index["clips"] = {k: list(v.shape) for k, v in clips.items()}
    index["clip_note"] = "uint8 [n,h,w,3]; frames = u8.astype(float32) / float32(255); 'big_half' is replicated 2x2 to 1024x576 first"
    (OUT / "index.json").write_text(json.dumps(index, indent=1))
    total = sum(p.stat().st_size for p in OUT.iterdir())
    print(f"wrote {len(list(OUT.iterdir()))} files, {total / 1e6:.1f} MB")


if __name__ == "__main__":
    main()
