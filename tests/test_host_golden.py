"""CPU tests: the host-side helpers and the motion_meta contract against golden vectors captured from
the reference's own NumPy helpers (tests/golden/make_golden.py, run in the build container)."""

import json
from pathlib import Path

import numpy as np
import pytest

GOLD = json.loads((Path(__file__).parent / "golden" / "reference_helpers.json").read_text())


def nd(x):
    return np.array(x["__nd__"], dtype=x["dtype"]).reshape(x["shape"])


def unwrap(x):
    if isinstance(x, dict) and "__nd__" in x:
        return nd(x)
    if isinstance(x, dict):
        return {k: unwrap(v) for k, v in x.items()}
    if isinstance(x, list):
        return [unwrap(v) for v in x]
    return x


@pytest.fixture(scope="module")
def hm(pkg):
    from vstab_amd import host_math

    return host_math


@pytest.fixture(scope="module")
def mv(pkg):
    from vstab_amd import meta_v2

    return meta_v2


def test_working_size(hm):
    for case in GOLD["working_size"]:
        out = hm._working_estimation_size(*case["in"])
        assert (list(out) if out is not None else None) == case["out"]


def test_params_roundtrip_and_rescale(hm):
    for case in GOLD["params"]:
        m = nd(case["matrix"])
        p = hm._matrix_to_params(m, case["mode"])
        assert p.dtype == np.float64 and np.array_equal(p, nd(case["params"]))
        back = hm._params_to_matrix(p, case["mode"])
        assert back.dtype == np.float32 and np.array_equal(back, nd(case["back"]))
        assert np.array_equal(hm._rescale_transform_to_full(m, (1920, 1080), (960, 540)), nd(case["rescaled"]))
        assert np.array_equal(hm._rescale_transform_to_full(m, (3840, 2160), (960, 540)), nd(case["rescaled_4k"]))


def test_smoothing_window_matches_reference_outputs(hm):
    """The window length is implied by the reference outputs: a constant-1 path stays 1, and the
    number of distinct rows affected near an impulse equals the window."""
    assert [hm.smoothing_window(s, 16.0) for s in (0.0, 0.5, 1.0)] == [3, 9, 13]
    assert [hm.smoothing_window(s, 24.0) for s in (0.0, 0.5, 1.0)] == [5, 13, 21]
    assert [hm.smoothing_window(s, 30.0) for s in (0.0, 0.5, 1.0)] == [7, 15, 25]
    assert [hm.smoothing_window(s, 60.0) for s in (0.0, 0.5, 1.0)] == [11, 31, 49]


def test_bbox_expand_ratio(hm):
    shake = GOLD["shake_small"]
    mats = [np.asarray(e["matrix"], dtype=np.float64) for e in shake["per_frame"]]
    mins, maxs = hm._compute_bounding_boxes(mats, 192, 108)
    b = GOLD["bbox"]
    assert np.array_equal(mins, nd(b["mins"])) and np.array_equal(maxs, nd(b["maxs"]))
    assert hm._min_content_ratio(mins, maxs, 192, 108) == b["ratio"]
    t, size = hm._prepare_expand_transform(mins, maxs)
    assert np.array_equal(t, nd(b["expand_matrix"])) and list(size) == b["expand_size"]
    lit = GOLD["expand_literal"]
    t, size = hm._prepare_expand_transform(nd(lit["mins"]), nd(lit["maxs"]))
    assert np.array_equal(t, nd(lit["matrix"])) and list(size) == lit["size"]
    assert hm._min_content_ratio(nd(lit["mins"]), nd(lit["maxs"]), 73, 45) == lit["ratio"]


def test_expand_matrices_and_blur_samples(pkg, hm):
    from vstab_amd import apply_pipeline as ap

    shake = GOLD["shake_small"]
    mats = [np.asarray(e["matrix"], dtype=np.float64) for e in shake["per_frame"]]
    em, size = ap._expand_matrices(mats, (192, 108))
    assert np.array_equal(np.stack(em), nd(GOLD["expand_matrices"]["matrices"]))
    assert list(size) == GOLD["expand_matrices"]["size"]
    # A4: the shipped sample-matrix routine is the library's (vstab_blur_sample_matrices, host arithmetic -- callable
    # without a GPU); the reference forms the samples in f64 (motion_apply.py:125-134) and casts each to f32 right
    # before cv2.warpPerspective (motion_apply.py:172): bit-exact after that cast, for every frame position the
    # golden holds (first / interior / last frame of the clip, single-frame clip)
    from vstab_amd import native

    for case in GOLD["blur_samples"]:
        got = native.blur_sample_matrices(np.stack(mats), case["idx"], 1, case["blur"], case["samples"])[0]
        assert got.dtype == np.float32 and np.array_equal(got, nd(case["out"]).astype(np.float32)), case
    whole = native.blur_sample_matrices(np.stack(mats), 0, len(mats), 0.5, 17)      # all frames in one call == per frame
    for case in GOLD["blur_samples"]:
        if case["samples"] == 17:
            assert np.array_equal(whole[case["idx"]], nd(case["out"]).astype(np.float32))
    single = native.blur_sample_matrices(np.stack(mats[:1]), 0, 1, 0.5, 9)
    assert single.shape == (1, 1, 3, 3) and np.array_equal(single[0], nd(GOLD["blur_single"]).astype(np.float32))


def test_padding_color(hm):
    for case in GOLD["padding_color"]:
        assert list(hm._parse_padding_color(case["in"])) == case["out"], case["in"]
    assert hm.border_value((127, 127, 127))[0] == np.float32(0.49803921580314636)


def test_warp_meta_and_motion_meta_blocks(hm, mv):
    shake = GOLD["shake_small"]
    mats = [np.asarray(e["matrix"], dtype=np.float64).astype(np.float32) for e in shake["per_frame"]][:3]
    warp = hm._build_stabilization_warp_meta(source_size=(192, 108), output_size=(200, 120), framing_mode="expand",
                                             applied_matrices=mats)
    assert warp == GOLD["warp_meta"]
    assert mv.applied_motion_meta_from_stabilization_warp(warp, fps=16.0, source="estimated_flow") == GOLD["mm_applied"]
    assert mv.motion_meta_from_stabilization_warp(warp, fps=24.0, source="legacy_stabilization") == GOLD["mm_inverse"]
    r = mv.resolve_motion_meta({"stabilization_warp": warp})
    g = GOLD["mm_resolve_legacy"]
    assert (r.source, r.frame_count, r.fps, list(r.input_size), list(r.output_size)) == (
        g["source"], g["frame_count"], g["fps"], g["input_size"], g["output_size"])
    assert np.array_equal(np.stack([t.matrix for t in r.per_frame]), nd(g["matrices"]))
    # generated shake blocks validate (generator present)
    mv.validate_motion_meta(shake)


def test_motion_meta_error_messages(mv):
    ok = GOLD["mm_ok"]

    def mutate(**kw):
        b = json.loads(json.dumps(ok))
        b.update(kw)
        return b

    nan = float("nan")
    calls = {
        "not_dict": lambda: mv.validate_motion_meta([]),
        "version": lambda: mv.validate_motion_meta(mutate(version=1)),
        "convention": lambda: mv.validate_motion_meta(mutate(matrix_convention="x")),
        "source": lambda: mv.validate_motion_meta(mutate(source="")),
        "frame_count_type": lambda: mv.validate_motion_meta(mutate(frame_count="abc")),
        "frame_count_neg": lambda: mv.validate_motion_meta(mutate(frame_count=-1)),
        "fps": lambda: mv.validate_motion_meta(mutate(fps=0)),
        "fps_type": lambda: mv.validate_motion_meta(mutate(fps="q")),
        "input_size": lambda: mv.validate_motion_meta(mutate(input_size=[1])),
        "input_size_neg": lambda: mv.validate_motion_meta(mutate(input_size=[0, 4])),
        "input_size_type": lambda: mv.validate_motion_meta(mutate(output_size=["a", 4])),
        "per_frame_type": lambda: mv.validate_motion_meta(mutate(per_frame={})),
        "count_mismatch": lambda: mv.validate_motion_meta(mutate(frame_count=3)),
        "entry_type": lambda: mv.validate_motion_meta(mutate(per_frame=[1, 2])),
        "entry_index": lambda: mv.validate_motion_meta(mutate(per_frame=[ok["per_frame"][1], ok["per_frame"][0]])),
        "entry_missing": lambda: mv.validate_motion_meta(mutate(per_frame=[{"index": 0}, ok["per_frame"][1]])),
        "entry_shape": lambda: mv.validate_motion_meta(mutate(per_frame=[{"index": 0, "matrix": [[1, 0], [0, 1]]}, ok["per_frame"][1]])),
        "entry_nan": lambda: mv.validate_motion_meta(mutate(per_frame=[{"index": 0, "matrix": [[nan, 0, 0], [0, 1, 0], [0, 0, 1]]}, ok["per_frame"][1]])),
        "entry_singular": lambda: mv.validate_motion_meta(mutate(per_frame=[{"index": 0, "matrix": [[0, 0, 0]] * 3}, ok["per_frame"][1]])),
        "shake_generator": lambda: mv.validate_motion_meta(mutate(source="generated_shake")),
        "resolve_not_dict": lambda: mv.resolve_motion_meta(3),
        "resolve_empty": lambda: mv.resolve_motion_meta({}),
        "warp_not_dict": lambda: mv.motion_meta_from_stabilization_warp(3, 16.0, "x"),
        "warp_convention": lambda: mv.applied_motion_meta_from_stabilization_warp({"matrix_convention": "q"}, 16.0, "x"),
        "warp_per_frame": lambda: mv.applied_motion_meta_from_stabilization_warp(
            {"matrix_convention": "source_to_stabilized", "source_size": [4, 4], "output_size": [4, 4], "per_frame": 3}, 16.0, "x"),
    }
    assert set(calls) == set(GOLD["mm_errors"])
    for name, fn in calls.items():
        expected = GOLD["mm_errors"][name]
        assert expected is not None, name
        with pytest.raises(ValueError) as info:
            fn()
        assert str(info.value) == expected["message"], name
    assert mv.build_motion_meta_v2(source="manual", frame_count=2, fps=16.0, input_size=(8, 6), output_size=(8, 6),
                                   matrices=[np.eye(3), np.eye(3)]) == ok


def test_normalize_video_input_layouts(hm):
    import torch

    for name, case in GOLD["normalize"].items():
        value = unwrap(case["input"])
        c = hm._normalize_video_input(value)
        assert (c.width, c.height, c.channels, c.fps, c.template_kind) == (
            case["width"], case["height"], case["channels"], case["fps"], case["template_kind"]), name
        frames = np.stack([np.asarray(f) for f in c.frames])
        assert frames.dtype == np.float32 and np.array_equal(frames, nd(case["frames"])), name
        rec = hm._reconstruct_video(c.frames, c)
        payload = rec["frames"] if isinstance(rec, dict) else rec
        assert isinstance(payload, torch.Tensor) and payload.dtype == torch.float32
        assert np.array_equal(payload.numpy(), nd(case["frames"])), name
    # the ComfyUI IMAGE fast path (float32 BHWC tensor) gives the same frames, incl. per-frame /255 sniffing
    batch = nd(GOLD["normalize"]["batch"]["frames"])
    t = torch.from_numpy(np.tile(batch, (1, 2, 2, 1)).copy())
    t[1] *= 255.0
    c = hm._normalize_video_input(t)
    ref = hm._normalize_video_input([f for f in t.numpy()])
    # the sniff is owed until a pass over the pixels has seen the per-frame maxima (the GPU pipelines take them from the
    # gray kernel; here the host form of the same rule settles it)
    assert c.batch is not None and c.range_pending and c.batch.data_ptr() == t.data_ptr()
    assert hm.resolve_value_range(c) and not c.range_pending and c.adapter.value_range == "0_1"
    assert np.array_equal(c.batch.numpy(), np.stack(ref.frames)) and np.array_equal(np.stack(c.frames), np.stack(ref.frames))
    assert t[1].max() > 1.5, "the caller's tensor is never modified"
    plain = hm._normalize_video_input(torch.from_numpy(np.tile(batch, (1, 2, 2, 1)).copy()))
    assert not hm.resolve_value_range(plain) and plain.batch is not None
    for name, err in GOLD["normalize_errors"].items():
        with pytest.raises(ValueError) as info:
            hm._normalize_video_input([] if name == "empty" else {"x": 1})
        assert str(info.value) == err["message"]


def test_batched_host_math_equals_per_item(hm):
    """The vectorised forms used on the hot path give the same bits as the reference-shaped per-item helpers."""
    rng = np.random.default_rng(3)
    n = 300
    th, sc = rng.uniform(-0.05, 0.05, n), rng.uniform(0.95, 1.05, n)
    stack = np.tile(np.eye(3, dtype=np.float32), (n, 1, 1))
    stack[:, 0, 0] = sc * np.cos(th); stack[:, 0, 1] = -sc * np.sin(th); stack[:, 1, 0] = sc * np.sin(th); stack[:, 1, 1] = sc * np.cos(th)
    stack[:, :2, 2] = rng.uniform(-40, 40, (n, 2))
    stack[::3, 2, :2] = rng.uniform(-1e-4, 1e-4, (n // 3, 2))
    for src, work in (((1920, 1080), (960, 540)), ((3840, 2160), (960, 540)), ((1280, 720), (960, 540)), ((1000, 777), (960, 746))):
        one = np.stack([hm._rescale_transform_to_full(m, src, work) for m in stack])
        assert np.array_equal(hm.rescale_transforms_to_full(stack, src, work), one)
    mins, maxs = hm._compute_bounding_boxes(list(stack), 1920, 1080)
    bmins, bmaxs = hm.bounding_boxes_batched(stack, 1920, 1080)
    assert np.array_equal(mins, bmins) and np.array_equal(maxs, bmaxs)
    shift = np.array([[1, 0, 3.25], [0, 1, -7.5], [0, 0, 1]], np.float32)
    assert np.array_equal(np.matmul(shift, stack), np.stack([shift @ m for m in stack]))
    # Motion Apply's matrices are float64 (parsed from the meta, or the inverse of a recorded warp): the expand geometry of
    # apply_pipeline._expand_matrices on such a stack, batched against per item
    s64 = stack.astype(np.float64) + rng.uniform(-1e-9, 1e-9, stack.shape)
    s64[:, 2, 2] = 1.0
    mins, maxs = hm._compute_bounding_boxes(list(s64), 3840, 2160)
    bmins, bmaxs = hm.bounding_boxes_batched(s64, 3840, 2160)
    assert np.array_equal(mins, bmins) and np.array_equal(maxs, bmaxs)
    sh, size = hm._prepare_expand_transform(mins, maxs)
    assert np.array_equal(np.matmul(sh, s64), np.stack([sh @ m for m in s64])) and np.matmul(sh, s64).dtype == np.float64
    from vstab_amd import apply_pipeline as ap

    out, out_size = ap._expand_matrices(list(s64), (3840, 2160))
    assert out_size == size and np.array_equal(np.stack(out), np.stack([sh @ m for m in s64]))
    for mode in ("translation", "similarity", "perspective"):
        per_item = np.stack([hm._matrix_to_params(m, mode) for m in stack])
        batch = hm.matrices_to_params(stack, mode)
        assert batch.dtype == np.float64 and np.array_equal(batch, per_item), mode
        back_item = np.stack([hm._params_to_matrix(p * 0.37, mode) for p in per_item])
        back = hm.params_to_matrices(per_item * 0.37, mode)
        assert back.dtype == np.float32 and np.array_equal(back, back_item), mode


def test_input_errors_and_soft_paths(hm):
    """Mixed layouts raise (stabilizer_utils.py:179-180); >3 channels are truncated, gray is repeated."""
    import torch

    a = np.zeros((6, 8, 3), np.float32)
    with pytest.raises(ValueError, match="Mixed tensor layouts within the same video sequence are not supported."):
        hm._normalize_video_input([a, torch.zeros((6, 8, 3))])
    with pytest.raises(ValueError, match="Mixed tensor layouts"):
        hm._normalize_video_input([a, np.zeros((3, 6, 8), np.float32)])
    c = hm._normalize_video_input([np.zeros((6, 8, 5), np.float32)])
    assert c.frames[0].shape == (6, 8, 3) and c.channels == 3
    c = hm._normalize_video_input([np.ones((6, 8), np.float32)])
    assert c.frames[0].shape == (6, 8, 3) and c.channels == 3


SMALL = json.loads((Path(__file__).parent / "golden" / "reference_small_paths.json").read_text())


@pytest.mark.parametrize("estimator", ["flow", "classic"])
def test_empty_and_single_frame_paths_match_reference(pkg, hm, estimator):
    """flow.py:242-310 / classic.py:189-250: the paths of `_stabilize_frames` that never reach a pixel kernel
    (no GPU needed).  The complete meta dict -- keys in order, values, types -- and the passthrough outputs
    equal what the reference's own function returned (captured by tests/golden/make_golden.py)."""
    from vstab_amd import flow_pipeline as fp

    one = nd(SMALL["single_frame"])
    for case in SMALL["cases"]:
        args = list(case["args"])
        args[6] = tuple(args[6])
        if case["kind"] == "single":
            value = {"frames": one, "fps": case["fps_in_dict"]} if case["fps_in_dict"] is not None else one
            context = hm._normalize_video_input(value)
        else:
            base = hm._normalize_video_input(one)
            context = hm.VideoContext([], base.adapter, 8, 6, 3, None, "sequence", {})
        res = fp._stabilize_frames(context, *args, estimator=estimator)
        want = case[estimator]
        assert list(res.meta.keys()) == list(want["meta"].keys()), (case["kind"], args)
        assert json.dumps(res.meta) == json.dumps(want["meta"]), (case["kind"], args)
        assert np.array_equal(np.asarray(res.frames, np.float32).reshape(nd(want["frames"]).shape), nd(want["frames"]))
        assert np.asarray(res.masks).shape == nd(want["masks"]).shape and not np.asarray(res.masks).any()


SHAKE = json.loads((Path(__file__).parent / "golden" / "shake_cases.json").read_text())


def test_shake_generator_matches_reference(pkg):
    """the shake generator's components and complete motion_meta blocks equal the reference's
    (shake_noise.py, run by tests/golden/make_golden.py) bit for bit -- same draws from default_rng(seed), same
    floating-point operation order -- for every style, a manual recipe with jitter + walking step, clamped
    out-of-range inputs, and 0/1/3-frame clips."""
    from vstab_amd import shake_generator as sg

    for case in SHAKE["cases"]:
        recipe = sg.ShakeRecipe(**case["recipe"])
        kw = dict(recipe=recipe, frame_count=case["frame_count"], fps=case["fps"], amount=case["amount"], speed=case["speed"],
                  seed=case["seed"])
        comp = sg.generate_shake_components(**kw)
        for name, want in case["components"].items():
            assert np.array_equal(comp[name], nd(want)), (case["style"], case["seed"], name)
        w, h = case["size"]
        blk = sg.generate_shake_motion_meta(width=w, height=h, style=case["style"],
                                            node="shake_generator" if case["style"] != "manual" else "shake_generator_manual", **kw)
        assert json.dumps(blk) == json.dumps(case["motion_meta"]), (case["style"], case["seed"])
    assert {k: tuple(v) for k, v in sg.STYLES.items()} == {c["style"]: tuple(c["recipe"].values()) for c in SHAKE["cases"][:5]}
    got = sg.recipe_to_dict(sg.recipe_from_mapping(SHAKE["mapping"]["in"]))
    assert got == SHAKE["mapping"]["out"]
    for key, kwargs in (("negative_frames", dict(frame_count=-1, width=8, height=8)), ("zero_width", dict(frame_count=2, width=0, height=8))):
        with pytest.raises(ValueError) as err:
            sg.generate_shake_motion_meta(recipe=sg.STYLES["handheld"], fps=16.0, amount=1.0, speed=1.0, seed=0, **kwargs)
        assert str(err.value) == SHAKE["errors"][key]["message"]


def _sticky_walk_reference(records, requested):
    """Per-pair loop in the shape of flow.py:324-339 + :156-210: try modes from the active one downwards, take the
    first accepted candidate, make its mode sticky; no candidate at all -> identity reported as translation."""
    order = {"perspective": ["perspective", "similarity", "translation"], "similarity": ["similarity", "translation"],
             "translation": ["translation"]}
    active, out = requested, []
    for rec in records:
        pick = None
        for mode in order[active]:
            cand = rec.get(mode)
            if cand is not None and cand["accepted"]:
                pick = (cand["matrix"], mode, cand["confidence"], cand["residual"])
                break
        if pick is None:
            pick = (np.eye(3, dtype=np.float32), "translation", 0.0, 0.0)
        if pick[1] != active:
            active = pick[1]
        out.append(pick)
    return out, active


@pytest.mark.parametrize("requested", ["translation", "similarity", "perspective"])
def test_sticky_mode_selection_equals_sequential_walk(pkg, requested):
    """`select_transitions` (run-length walk on the record table, with its all-accepted fast path) against a plain
    per-pair loop on random candidate tables: rejections, missing candidates, pairs with nothing computed."""
    from vstab_amd import flow_pipeline as fp

    rng = np.random.default_rng({"translation": 1, "similarity": 2, "perspective": 3}[requested])
    modes = ["translation", "similarity", "perspective"]
    for trial in range(60):
        pairs = int(rng.integers(1, 40))
        p_reject = [0.0, 0.05, 0.5][trial % 3]
        records = []
        for _ in range(pairs):
            rec = {}
            if rng.random() >= p_reject * 0.3:                       # otherwise: fewer than 12 valid samples, nothing computed
                for mi, name in enumerate(modes[: modes.index(requested) + 1]):
                    if name != "translation" and rng.random() < p_reject * 0.2:
                        continue                                      # estimator returned no model
                    rec[name] = {"matrix": rng.normal(0, 1, (3, 3)).astype(np.float32), "confidence": float(rng.random()),
                                 "residual": float(rng.random()), "accepted": bool(name == "translation" or rng.random() >= p_reject)}
            records.append(rec)
        want, want_active = _sticky_walk_reference(records, requested)
        mats, used, confs, resids, active = fp.select_transitions(records, requested)
        assert active == want_active and used == [w[1] for w in want]
        assert confs == [w[2] for w in want] and resids == [w[3] for w in want]
        assert np.array_equal(mats, np.stack([w[0] for w in want]))


def test_progress_replay_equals_per_item_loop(pkg):
    """flow.py:347-351 / 589-593: one update per 10 finished items plus one for the remainder."""
    from vstab_amd import flow_pipeline as fp

    class Bar:
        def __init__(self):
            self.calls = []

        def update_absolute(self, value, total):
            self.calls.append((value, total))

    for count in list(range(0, 35)) + [99, 100, 101, 255, 256]:
        for done0 in (0, 7):
            want, pending, done = [], 0, done0
            for idx in range(count):
                pending += 1
                if pending >= 10 or idx == count - 1:
                    done += pending
                    want.append((done, 1000))
                    pending = 0
            bar = Bar()
            assert fp._replay_progress(bar, done0, count, 1000) == done0 + count and bar.calls == want, count
