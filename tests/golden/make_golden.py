#!/usr/bin/env python3
"""Generate golden vectors by running the REFERENCE's own pure-NumPy helpers in this container.

Run here only (needs /root/reference); the committed output (tests/golden/*.json) is data --
inputs and expected outputs -- and is what travels to the GPU box.  The reference's modules that
import cv2 are loaded under an empty placeholder module (cv2 is absent from this image; the same
trick the reference's scripts use for `comfy`, scripts/compare_refactor_behavior.py:75-109): every
function called below is NumPy-only, nothing that reaches a cv2.* call can run.

    python tests/golden/make_golden.py
"""

from __future__ import annotations

import json
import sys
import types
from pathlib import Path

import numpy as np

REF = Path("/root/reference")
OUT = Path(__file__).resolve().parent


def _install_placeholders() -> None:
    sys.modules.setdefault("cv2", types.ModuleType("cv2"))
    comfy = types.ModuleType("comfy")
    comfy_utils = types.ModuleType("comfy.utils")
    comfy_utils.ProgressBar = type("ProgressBar", (), {"__init__": lambda s, t: None, "update_absolute": lambda s, *a: None})
    comfy.utils = comfy_utils
    sys.modules.setdefault("comfy", comfy)
    sys.modules.setdefault("comfy.utils", comfy_utils)
    api = types.ModuleType("comfy_api")
    latest = types.ModuleType("comfy_api.latest")
    latest.ComfyExtension = type("ComfyExtension", (), {})
    latest.io = types.SimpleNamespace(Custom=lambda k: types.SimpleNamespace(Input=lambda *a, **k: None, Output=lambda *a, **k: None),
                                      ComfyNode=object)
    api.latest = latest
    sys.modules.setdefault("comfy_api", api)
    sys.modules.setdefault("comfy_api.latest", latest)


def jsonable(x):
    if isinstance(x, np.ndarray):
        return {"__nd__": x.tolist(), "dtype": str(x.dtype), "shape": list(x.shape)}
    if isinstance(x, (np.floating,)):
        return float(x)
    if isinstance(x, (np.integer,)):
        return int(x)
    if isinstance(x, dict):
        return {str(k): jsonable(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [jsonable(v) for v in x]
    return x


def error_of(fn, *a, **k):
    try:
        fn(*a, **k)
    except Exception as exc:  # noqa: BLE001
        return {"type": type(exc).__name__, "message": str(exc)}
    return None


def main() -> None:
    _install_placeholders()
    sys.path.insert(0, str(REF))
    from nodes import motion_meta as mm  # noqa: E402
    from nodes import shake_noise as sn  # noqa: E402
    from nodes import stabilizer_utils as su  # noqa: E402
    from nodes import motion_apply as ma  # noqa: E402

    g: dict = {}

    # ---- host helpers (F1, F6-F12, A4, A6) -------------------------------------------------
    g["working_size"] = [{"in": [w, h], "out": su._working_estimation_size(w, h)}
                         for (w, h) in [(1920, 1080), (3840, 2160), (854, 480), (640, 480), (960, 540), (1280, 720),
                                        (1080, 1920), (961, 400), (2000, 3)]]
    mats = {
        "translation": np.array([[1.0, 0.0, 2.5], [0.0, 1.0, -1.25], [0.0, 0.0, 1.0]], dtype=np.float32),
        "similarity": np.array([[1.02, -0.03, 2.0], [0.03, 1.02, -3.0], [0.0, 0.0, 1.0]], dtype=np.float32),
        "perspective": np.array([[1.01, 0.02, 2.0], [-0.01, 0.99, -1.0], [0.0002, -0.0001, 1.0]], dtype=np.float32),
    }
    g["params"] = []
    for mode, m in mats.items():
        p = su._matrix_to_params(m, mode)
        g["params"].append({"mode": mode, "matrix": m, "params": p, "back": su._params_to_matrix(p, mode),
                            "rescaled": su._rescale_transform_to_full(m, (1920, 1080), (960, 540)),
                            "rescaled_4k": su._rescale_transform_to_full(m, (3840, 2160), (960, 540))})
    rng = np.random.default_rng(7)
    walk = np.cumsum(rng.normal(0, 1.5, (40, 4)), axis=0)
    lin = np.stack([np.linspace(0.0, 4.0, 8), np.linspace(1.0, -2.0, 8), np.sin(np.linspace(0.0, 1.5, 8)),
                    np.cos(np.linspace(0.0, 1.5, 8))], axis=1)
    g["smooth"] = []
    for name, path in (("lin8", lin), ("walk40", walk), ("short2", lin[:2])):
        for smooth in (0.0, 0.25, 0.5, 1.0):
            for fps in (1.0, 16.0, 24.0, 30.0, 60.0):
                g["smooth"].append({"path": name, "smooth": smooth, "fps": fps, "out": su._smooth_path(path, smooth, fps)})
    g["smooth_paths"] = {"lin8": lin, "walk40": walk, "short2": lin[:2]}

    shake = sn.generate_shake_motion_meta(recipe=sn.STYLES["handheld"], frame_count=24, width=192, height=108, fps=16.0,
                                          amount=1.0, speed=1.0, seed=0, node="shake_generator", style="handheld")
    g["shake_small"] = shake
    smats = [np.asarray(e["matrix"], dtype=np.float64) for e in shake["per_frame"]]
    mins, maxs = su._compute_bounding_boxes(smats, 192, 108)
    t, size = su._prepare_expand_transform(mins, maxs)
    g["bbox"] = {"mins": mins, "maxs": maxs, "ratio": su._min_content_ratio(mins, maxs, 192, 108),
                 "expand_matrix": t, "expand_size": list(size)}
    emats, esize = ma._expand_matrices(smats, (192, 108))
    g["expand_matrices"] = {"matrices": np.stack(emats), "size": list(esize)}
    mins2 = np.array([[-2.0, 1.0], [0.5, -3.0], [1.5, 0.0]], dtype=np.float32)
    maxs2 = np.array([[73.5, 47.0], [75.0, 45.5], [72.0, 49.0]], dtype=np.float32)
    t2, s2 = su._prepare_expand_transform(mins2, maxs2)
    g["expand_literal"] = {"mins": mins2, "maxs": maxs2, "matrix": t2, "size": list(s2),
                           "ratio": su._min_content_ratio(mins2, maxs2, 73, 45)}
    g["padding_color"] = [{"in": v, "out": list(su._parse_padding_color(v))}
                          for v in ["#7F7F7F", "#404040", "#abc", "fff", "12,34,56", "300/0/-4", "7", "7,", "junk", "#12345",
                                    "  #00ff80 ", 0x102030, -5, 2 ** 30, "1,2", "1 2 3,"]]
    g["blur_samples"] = []
    for s in (3, 5, 9, 17, 33):
        for idx in (0, 11, 23):
            g["blur_samples"].append({"samples": s, "idx": idx, "blur": 0.5,
                                      "out": np.stack(ma._blurred_matrix_samples(smats, idx, 0.5, s))})
    g["blur_single"] = np.stack(ma._blurred_matrix_samples(smats[:1], 0, 0.5, 9))
    g["warp_meta"] = su._build_stabilization_warp_meta(source_size=(192, 108), output_size=(200, 120), framing_mode="expand",
                                                       applied_matrices=[m.astype(np.float32) for m in smats[:3]])

    # ---- motion_meta contract (F14, A1) ------------------------------------------------------
    warp = g["warp_meta"]
    g["mm_applied"] = mm.applied_motion_meta_from_stabilization_warp(warp, fps=16.0, source="estimated_flow")
    g["mm_inverse"] = mm.motion_meta_from_stabilization_warp(warp, fps=24.0, source="legacy_stabilization")
    r = mm.resolve_motion_meta({"stabilization_warp": warp})
    g["mm_resolve_legacy"] = {"source": r.source, "frame_count": r.frame_count, "fps": r.fps, "input_size": list(r.input_size),
                              "output_size": list(r.output_size), "matrices": np.stack([t.matrix for t in r.per_frame])}
    bad_cases = {}
    ok = mm.build_motion_meta_v2(source="manual", frame_count=2, fps=16.0, input_size=(8, 6), output_size=(8, 6),
                                 matrices=[np.eye(3), np.eye(3)])
    def mutate(**kw):
        b = json.loads(json.dumps(ok))
        b.update(kw)
        return b
    bad_cases["not_dict"] = error_of(mm.validate_motion_meta, [])
    bad_cases["version"] = error_of(mm.validate_motion_meta, mutate(version=1))
    bad_cases["convention"] = error_of(mm.validate_motion_meta, mutate(matrix_convention="x"))
    bad_cases["source"] = error_of(mm.validate_motion_meta, mutate(source=""))
    bad_cases["frame_count_type"] = error_of(mm.validate_motion_meta, mutate(frame_count="abc"))
    bad_cases["frame_count_neg"] = error_of(mm.validate_motion_meta, mutate(frame_count=-1))
    bad_cases["fps"] = error_of(mm.validate_motion_meta, mutate(fps=0))
    bad_cases["fps_type"] = error_of(mm.validate_motion_meta, mutate(fps="q"))
    bad_cases["input_size"] = error_of(mm.validate_motion_meta, mutate(input_size=[1]))
    bad_cases["input_size_neg"] = error_of(mm.validate_motion_meta, mutate(input_size=[0, 4]))
    bad_cases["input_size_type"] = error_of(mm.validate_motion_meta, mutate(output_size=["a", 4]))
    bad_cases["per_frame_type"] = error_of(mm.validate_motion_meta, mutate(per_frame={}))
    bad_cases["count_mismatch"] = error_of(mm.validate_motion_meta, mutate(frame_count=3))
    bad_cases["entry_type"] = error_of(mm.validate_motion_meta, mutate(per_frame=[1, 2]))
    bad_cases["entry_index"] = error_of(mm.validate_motion_meta, mutate(per_frame=[ok["per_frame"][1], ok["per_frame"][0]]))
    bad_cases["entry_missing"] = error_of(mm.validate_motion_meta, mutate(per_frame=[{"index": 0}, ok["per_frame"][1]]))
    bad_cases["entry_shape"] = error_of(mm.validate_motion_meta, mutate(per_frame=[{"index": 0, "matrix": [[1, 0], [0, 1]]}, ok["per_frame"][1]]))
    bad_cases["entry_nan"] = error_of(mm.validate_motion_meta, mutate(per_frame=[{"index": 0, "matrix": [[float("nan"), 0, 0], [0, 1, 0], [0, 0, 1]]}, ok["per_frame"][1]]))
    bad_cases["entry_singular"] = error_of(mm.validate_motion_meta, mutate(per_frame=[{"index": 0, "matrix": [[0, 0, 0], [0, 0, 0], [0, 0, 0]]}, ok["per_frame"][1]]))
    bad_cases["shake_generator"] = error_of(mm.validate_motion_meta, mutate(source="generated_shake"))
    bad_cases["resolve_not_dict"] = error_of(mm.resolve_motion_meta, 3)
    bad_cases["resolve_empty"] = error_of(mm.resolve_motion_meta, {})
    bad_cases["warp_not_dict"] = error_of(mm.motion_meta_from_stabilization_warp, 3, 16.0, "x")
    bad_cases["warp_convention"] = error_of(mm.applied_motion_meta_from_stabilization_warp, {"matrix_convention": "q"}, 16.0, "x")
    bad_cases["warp_per_frame"] = error_of(mm.applied_motion_meta_from_stabilization_warp,
                                           {"matrix_convention": "source_to_stabilized", "source_size": [4, 4], "output_size": [4, 4], "per_frame": 3}, 16.0, "x")
    g["mm_ok"] = ok
    g["mm_errors"] = bad_cases

    # ---- input adaptation (F0) -----------------------------------------------------------------
    yy, xx = np.mgrid[0:6, 0:8]
    base = [np.stack([(xx + i) / 12.0, yy / 7.0, ((xx + yy + i) % 5) / 4.0], -1).astype(np.float32) for i in range(3)]
    batch = np.stack(base, 0)
    layouts = {
        "list": base,
        "batch": batch,
        "dict": {"frames": batch, "fps": 24.0},
        "wrapped": [f[np.newaxis] for f in base],
        "float64": batch.astype(np.float64),
        "uint8": (batch * 255.0).round().clip(0, 255).astype(np.uint8),
        "float255": (batch * 255.0).astype(np.float32),
        "chw": [np.moveaxis(f, -1, 0) for f in base],
        "gray": [f[..., :1] for f in base],
        "rgba": [np.concatenate([f, f[..., :1]], -1) for f in base],
    }
    g["normalize"] = {}
    for name, value in layouts.items():
        c = su._normalize_video_input(value)
        g["normalize"][name] = {"input": value if not isinstance(value, dict) else {"frames": value["frames"], "fps": value["fps"]},
                                "width": c.width, "height": c.height, "channels": c.channels, "fps": c.fps,
                                "template_kind": c.template_kind, "frames": np.stack(c.frames)}
    g["normalize_errors"] = {"empty": error_of(su._normalize_video_input, []),
                             "dict_missing": error_of(su._normalize_video_input, {"x": 1})}

    (OUT / "reference_helpers.json").write_text(json.dumps(jsonable(g)))

    # ---- the node pipelines' paths that never reach a cv2 call: empty and single-frame clips -----------
    # (flow.py:242-310, classic.py:189-250).  Full meta dicts of the reference for Flow and Classic.
    from nodes import video_stabilizer_classic as vc  # noqa: E402
    from nodes import video_stabilizer_flow as vf  # noqa: E402

    small: dict = {"cases": []}
    one = (np.arange(6 * 8 * 3, dtype=np.float32).reshape(1, 6, 8, 3) % 17) / 16.0
    calls = [
        ("single", one, ("crop_and_pad", "similarity", False, 0.7, 0.5, 0.6, (127, 127, 127), 16.0)),
        ("single", one, ("expand", "perspective", True, 0.3, 0.2, 1.0, (1, 2, 3), 0.0)),
        ("single", {"frames": one, "fps": 24.0}, ("crop", "translation", False, 1.0, 0.0, 0.6, (255, 0, 16), -5.0)),
        ("empty", None, ("crop_and_pad", "similarity", False, 0.7, 0.5, 0.6, (127, 127, 127), 16.0)),
        ("empty", None, ("expand", "perspective", True, 0.3, 0.2, 0.6, (1, 2, 3), 0.0)),
    ]
    for kind, value, args in calls:
        if kind == "single":
            ctx = su._normalize_video_input(value)
        else:
            base_ctx = su._normalize_video_input(one)
            ctx = su.VideoContext([], base_ctx.adapter, 8, 6, 3, None, "sequence", {})
        entry = {"kind": kind, "args": list(args), "fps_in_dict": value["fps"] if isinstance(value, dict) else None}
        for name, mod in (("flow", vf), ("classic", vc)):
            r = mod._stabilize_frames(ctx, *args)
            entry[name] = {"meta": r.meta, "frames": np.asarray(r.frames), "masks": np.asarray(r.masks)}
        small["cases"].append(entry)
    small["single_frame"] = one
    (OUT / "reference_small_paths.json").write_text(json.dumps(jsonable(small)))

    # ---- shake generator: full motion_meta blocks + raw components for every style and a manual
    # recipe that exercises the jitter and walking-step layers; RNG draw order is part of the contract
    shake: dict = {"cases": [], "errors": {}}
    manual = sn.ShakeRecipe(pan=0.7, tilt=0.2, roll=1.3, zoom=0.01, drift_freq=0.6, tremor=0.9, tremor_freq=7.5, jitter_rate=2.0,
                            step=1.1, randomness=0.8, virtual_fov=35.0)
    wild = sn.ShakeRecipe(pan=9.0, tilt=-1.0, roll=0.0, zoom=1.0, drift_freq=0.0, tremor=3.0, tremor_freq=0.2, jitter_rate=9.0,
                          step=0.0, randomness=0.0, virtual_fov=500.0)    # clamped on entry
    plans = [(name, sn.STYLES[name], 24, (640, 360), 16.0, 1.0, 1.0, 0) for name in sn.STYLES]
    plans += [("handheld", sn.STYLES["handheld"], 40, (1920, 1080), 24.0, 2.5, 0.4, 123456789),
              ("walking", sn.STYLES["walking"], 33, (1080, 1920), 30.0, 0.5, 3.0, 7),
              ("action", sn.STYLES["action"], 48, (854, 480), 12.0, 1.0, 1.0, 2 ** 40 + 3),
              ("manual", manual, 36, (720, 576), 25.0, 1.5, 1.2, 99),
              ("manual", wild, 20, (320, 240), 0.5, 9.0, 0.01, 5),
              ("manual", manual, 1, (64, 48), 16.0, 1.0, 1.0, 1),
              ("manual", manual, 0, (64, 48), 16.0, 1.0, 1.0, 1),
              ("manual", manual, 3, (64, 48), 16.0, 1.0, 1.0, 1)]
    for style, recipe, n, (w, h), fps, amount, speed, seed in plans:
        comp = sn.generate_shake_components(recipe=recipe, frame_count=n, fps=fps, amount=amount, speed=speed, seed=seed)
        blk = sn.generate_shake_motion_meta(recipe=recipe, frame_count=n, width=w, height=h, fps=fps, amount=amount, speed=speed,
                                            seed=seed, node="shake_generator" if style != "manual" else "shake_generator_manual",
                                            style=style)
        shake["cases"].append({"style": style, "recipe": sn.recipe_to_dict(recipe), "frame_count": n, "size": [w, h], "fps": fps,
                               "amount": amount, "speed": speed, "seed": seed,
                               "components": {"pan_deg": comp.pan_deg, "tilt_deg": comp.tilt_deg, "roll_deg": comp.roll_deg,
                                              "zoom_log": comp.zoom_log},
                               "motion_meta": blk})
    shake["errors"]["negative_frames"] = error_of(sn.generate_shake_motion_meta, recipe=manual, frame_count=-1, width=8, height=8,
                                                  fps=16.0, amount=1.0, speed=1.0, seed=0)
    shake["errors"]["zero_width"] = error_of(sn.generate_shake_motion_meta, recipe=manual, frame_count=2, width=0, height=8,
                                             fps=16.0, amount=1.0, speed=1.0, seed=0)
    shake["mapping"] = {"in": {k: v * 10 for k, v in sn.recipe_to_dict(manual).items()},
                        "out": sn.recipe_to_dict(sn.recipe_from_mapping({k: v * 10 for k, v in sn.recipe_to_dict(manual).items()}))}
    (OUT / "shake_cases.json").write_text(json.dumps(jsonable(shake)))

    # larger shake clips used as Motion Apply inputs for configs C3/C5 (matrices only)
    for tag, (n, w, h) in {"c3_256x1080p": (256, 1920, 1080), "c5_64x4k": (64, 3840, 2160)}.items():
        blk = sn.generate_shake_motion_meta(recipe=sn.STYLES["handheld"], frame_count=n, width=w, height=h, fps=16.0,
                                            amount=1.0, speed=1.0, seed=0, node="shake_generator", style="handheld")
        (OUT / f"shake_{tag}.json").write_text(json.dumps(blk))
    print("wrote", [p.name for p in OUT.glob("*.json")])


if __name__ == "__main__":
    main()
