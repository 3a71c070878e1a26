"""The plan formed on the device (csrc/vstab_traj.hip: plan_kernel) against the host plan (the reference's arithmetic with
the host's libm), and the speculation built on it (flow_pipeline._stabilize_with_device_plan): the Flow node returns the
host plan's result bit for bit whether the device plan is used or not, also when the device plan is (made) wrong."""

import numpy as np
import pytest

import bench
from tests.util import shake_path

pytestmark = pytest.mark.gpu


def _clip(ctx, n, w, h, kind, amp=1.0, seed=3):
    cam = shake_path(n, w, h, kind, seed=seed, amp=amp)
    return bench.synth_clip(n, 0, h, w, ctx.device, mats=cam)


def _fits(ctx, frames, mode):
    from vstab_amd import flow_pipeline as fp
    from vstab_amd import host_math as hm

    n, h, w, _ = frames.shape
    work = hm._working_estimation_size(w, h)
    gray = ctx.gray_downscale(frames, work)
    _, grid = ctx.dis_flow_batch(gray, sample_step=fp.SAMPLE_STEP, want_full=False, want_grid=True)
    pairs = ctx.sample_fit_batch_begin(grid, fp.SAMPLE_STEP, mode)
    return pairs, work


@pytest.mark.parametrize("mode,size,camera_lock,smooth,strength,fps", [
    ("similarity", (1920, 1080), False, 0.5, 0.7, 16.0),      # C2's configuration: estimated at 960x540, rescaled
    ("similarity", (960, 540), False, 1.0, 1.0, 30.0),        # estimated at full size (no rescale), widest window
    ("similarity", (640, 360), True, 0.5, 0.7, 16.0),         # camera_lock: target path 0
    ("translation", (960, 540), False, 0.0, 0.4, 16.0),       # smooth 0: the path is its own target
    ("translation", (1280, 720), False, 0.5, 0.7, 24.0),
    ("perspective", (1920, 1080), False, 0.5, 0.7, 16.0),     # C3's Flow half: eight parameters, no libm; rescaled
    ("perspective", (960, 540), True, 1.0, 1.0, 30.0),
])
def test_device_plan_equals_host_plan(ctx, pkg, mode, size, camera_lock, smooth, strength, fps):
    from vstab_amd import flow_pipeline as fp

    w, h = size
    n = 24
    frames = _clip(ctx, n, w, h, mode, amp=1.5)
    pairs, work = _fits(ctx, frames, mode)
    ctx.flow_plan_device(ctx.fit_records_device(), pairs, mode, size, work, smooth, fps, strength, camera_lock)
    table = ctx.sample_fit_batch_end(pairs)
    final_dev, path_dev, target_dev, region = ctx.flow_plan_result(n, fp.PLAN_PARAMS[mode])
    plan = fp.plan_stabilization(ctx, table, size, n, "crop_and_pad", mode, camera_lock, strength, smooth, 0.6, (127, 127, 127),
                                 fps, fps)
    em = plan.estimated_motion
    # translation and perspective parameters involve no libm at all: everything is bit-equal; similarity: the path may differ
    # in the last unit of a double (device atan2 / log vs glibc), the float32 matrices must still agree
    if mode != "similarity":
        assert np.array_equal(path_dev, em["path"]) and np.array_equal(target_dev, em["target_path"])
    else:
        assert np.allclose(path_dev, em["path"], rtol=0, atol=4e-15 * max(1.0, np.abs(em["path"]).max()))
        assert np.allclose(target_dev, em["target_path"], rtol=0, atol=4e-15 * max(1.0, np.abs(em["path"]).max()))
    assert np.array_equal(final_dev.view(np.uint32), plan.final_matrices.view(np.uint32))
    fm = plan.framing_meta
    x0, y0 = fm["safe_region_origin"]
    assert region[0] == x0 and region[1] == y0


def test_host_trajectory_equals_the_plan_kernels(ctx, pkg):
    """vstab_trajectory (host since round 4) and plan_kernel share one operation order: on deltas that involve no libm
    (translation) the device path / target and the host's are the same bits."""
    from vstab_amd import flow_pipeline as fp
    from vstab_amd import native

    frames = _clip(ctx, 40, 960, 540, "translation", amp=2.0)
    pairs, work = _fits(ctx, frames, "translation")
    ctx.flow_plan_device(ctx.fit_records_device(), pairs, "translation", (960, 540), work, 0.5, 16.0, 0.7, False)
    table = ctx.sample_fit_batch_end(pairs)
    _, path_dev, target_dev, _ = ctx.flow_plan_result(40, 2)
    mats = fp.select_transitions(table, "translation")[0]
    _, deltas = native.transitions_to_params(mats, "translation", (960, 540), work)
    path, target = ctx.trajectory(deltas, 0.5, 16.0, 0.7, False)
    assert np.array_equal(path, path_dev) and np.array_equal(target, target_dev)


def test_host_trajectory_tracks_the_plan_kernel_on_similarity_deltas(ctx, pkg):
    """The same tie on SIMILARITY deltas (tx, ty, rotation, log-scale).  NOT bit-equal, and the assert says why: the deltas
    come from atan2 / log, which the device's fp64 library and the host's glibc round differently in the last unit of a
    double; prefix sum, box filter and blend then run in one operation order on both sides, so the difference stays at that
    size: <= 16 units in the last place of the largest path value over 40 frames (measured: 0-2 ulp).  The host's smoothing of
    the HOST's own deltas is pinned exactly by the reference goldens (test_fit_gpu.py); this test pins that the kernel runs
    the same filter on similarity-shaped data (window, edge clamping, blend), which the translation test above cannot see in
    the rotation / scale columns."""
    from vstab_amd import flow_pipeline as fp
    from vstab_amd import native

    n = 40
    frames = _clip(ctx, n, 960, 540, "similarity", amp=2.0)
    pairs, work = _fits(ctx, frames, "similarity")
    ctx.flow_plan_device(ctx.fit_records_device(), pairs, "similarity", (960, 540), work, 0.5, 16.0, 0.7, False)
    table = ctx.sample_fit_batch_end(pairs)
    _, path_dev, target_dev, _ = ctx.flow_plan_result(n, 4)
    mats = fp.select_transitions(table, "similarity")[0]
    _, deltas = native.transitions_to_params(mats, "similarity", (960, 540), work)
    path, target = ctx.trajectory(deltas, 0.5, 16.0, 0.7, False)
    assert path.shape == path_dev.shape == (n, 4) and np.abs(path[:, 2:]).max() > 1e-4   # rotation / scale columns are exercised
    for host, dev in ((path, path_dev), (target, target_dev)):
        for col in range(4):
            ulp = np.spacing(np.abs(host[:, col]).max())
            assert np.abs(host[:, col] - dev[:, col]).max() <= 16 * ulp, (col, np.abs(host[:, col] - dev[:, col]).max() / ulp)


def _run(ctx, frames, mode, monkeypatch, device_plan, framing="crop_and_pad"):
    from vstab_amd import flow_pipeline as fp
    from vstab_amd import host_math as hm

    monkeypatch.setenv("VSTAB_DEVICE_PLAN", "1" if device_plan else "0")
    res = fp._stabilize_frames(hm._normalize_video_input(frames), framing, mode, False, 0.7, 0.5, 0.6, (127, 127, 127), 16.0,
                               ctx=ctx, keep_on_device=True)
    return res, res.device_plan   # what the speculation did travels in the result, not in module state


@pytest.mark.parametrize("framing", ["crop_and_pad", "expand"])
@pytest.mark.parametrize("mode", ["similarity", "translation"])
def test_flow_node_is_the_same_with_and_without_the_device_plan(ctx, pkg, monkeypatch, mode, framing):
    """expand (round 5): the plan kernel shifts the frames' union to the origin and the host sizes the canvas from the kernel's
    region before it queues the warp; same pixels, same canvas, same meta as the host-plan flow."""
    frames = _clip(ctx, 16, 1280, 720, "similarity", amp=1.5)
    a, info_a = _run(ctx, frames, mode, monkeypatch, device_plan=False, framing=framing)
    b, info_b = _run(ctx, frames, mode, monkeypatch, device_plan=True, framing=framing)
    assert tuple(a.frames.shape) == tuple(b.frames.shape) and (framing == "crop_and_pad") == (tuple(a.frames.shape[1:3]) == (720, 1280))
    assert info_a == {"used": False, "mismatched_frames": 0} and info_b == {"used": True, "mismatched_frames": 0}
    assert a.meta == b.meta
    assert bool((a.frames == b.frames).all()) and bool((a.masks == b.masks).all())


@pytest.mark.parametrize("framing", ["crop_and_pad", "expand"])
def test_perspective_flow_is_the_same_with_and_without_the_device_plan(ctx, pkg, monkeypatch, framing):
    """C3's Flow half on the device plan: same pixels, masks and meta as the host-plan flow.  Frames whose final = T @ M the
    device rounded differently from NumPy's matmul (sums of two inexact terms) are warped again: a few at most."""
    frames = _clip(ctx, 16, 1280, 720, "perspective", amp=1.5)
    a, info_a = _run(ctx, frames, "perspective", monkeypatch, device_plan=False, framing=framing)
    b, info_b = _run(ctx, frames, "perspective", monkeypatch, device_plan=True, framing=framing)
    assert info_a == {"used": False, "mismatched_frames": 0} and info_b["used"] is True
    assert info_b["mismatched_frames"] <= (2 if framing == "crop_and_pad" else 16)   # (expand: a canvas one pixel off re-warps all)
    assert a.meta["transform_mode_applied"] == "perspective" and a.meta == b.meta
    assert tuple(a.frames.shape) == tuple(b.frames.shape)
    assert bool((a.frames == b.frames).all()) and bool((a.masks == b.masks).all())


def test_a_wrong_device_plan_is_caught_and_the_frame_warped_again(pkg):
    """The TEST build's VSTAB_DEBUG_PLAN_PERTURB makes the device plan's matrix of one frame wrong by one ulp: the host's
    verification must catch it, warp that frame again, and return the host plan's result (the shipped library has no such
    knob; a child process loads the test build)."""
    from tests.util import run_with_hooks

    run_with_hooks("""
        import bench
        from tests.util import shake_path
        from vstab_amd import flow_pipeline as fp, host_math as hm
        ctx = native.Context(0)
        frames = bench.synth_clip(12, 0, 540, 960, ctx.device, mats=shake_path(12, 960, 540, "similarity", seed=3, amp=1.5))

        def run(device_plan, perturb=None):
            os.environ["VSTAB_DEVICE_PLAN"] = "1" if device_plan else "0"
            os.environ.pop("VSTAB_DEBUG_PLAN_PERTURB", None)
            if perturb is not None:
                os.environ["VSTAB_DEBUG_PLAN_PERTURB"] = str(perturb)
            return fp._stabilize_frames(hm._normalize_video_input(frames), "crop_and_pad", "similarity", False, 0.7, 0.5, 0.6,
                                        (127, 127, 127), 16.0, ctx=ctx, keep_on_device=True)

        want = run(False)
        got = run(True, perturb=5)
        assert got.device_plan == {"used": True, "mismatched_frames": 1}, got.device_plan
        assert want.device_plan == {"used": False, "mismatched_frames": 0}
        assert want.meta == got.meta
        assert bool((want.frames == got.frames).all()) and bool((want.masks == got.masks).all())
    """)


def test_value_range_rescale_discards_the_speculative_run(ctx, pkg, monkeypatch):
    """F0: 0..255 float frames are found by the sniff that rides on the gray pass; the speculative warp used the unscaled
    frames and must not survive."""
    import torch

    frames = _clip(ctx, 6, 640, 360, "similarity") * 255.0
    a, info_a = _run(ctx, frames.clone(), "similarity", monkeypatch, device_plan=False)
    b, info_b = _run(ctx, frames.clone(), "similarity", monkeypatch, device_plan=True)
    assert info_b["used"] is False
    assert a.meta == b.meta and bool((a.frames == b.frames).all())
    assert float(b.frames.max()) <= 1.0 + 1e-6 and torch.isfinite(b.frames).all()


def test_plan_kernel_refuses_what_it_does_not_cover(ctx, pkg):
    from vstab_amd import native

    frames = _clip(ctx, 4, 640, 360, "similarity")
    pairs, work = _fits(ctx, frames, "similarity")
    with pytest.raises(native.VstabError, match="unknown model"):
        native._check(ctx.lib.vstab_flow_plan_device(ctx.handle, ctx.fit_records_device(), pairs, 3, None, None, 0.5, 16.0, 0.7, 0, 640, 360,
                                                     0, None, 0, 0), "vstab_flow_plan_device")
    with pytest.raises(native.VstabError, match="framing"):
        ctx.flow_plan_device(ctx.fit_records_device(), pairs, "similarity", (640, 360), work, 0.5, 16.0, 0.7, False, framing="crop")
    ctx.sample_fit_batch_end(pairs)
    with pytest.raises(native.VstabError, match="no fit"):
        ctx.sample_fit_batch_end(pairs)


def test_sticky_mode_walk_of_a_perspective_plan(ctx, pkg):
    """Three models: plan_kernel replays flow.py:324-339 pair by pair from the usable-fit flags (perspective until its fit is
    rejected, then the best usable model below, which becomes the active one; none: identity, active = translation).  Tables
    with every kind of step -- 2 -> 1 -> 0, 2 -> 0 directly, a pair with nothing usable, rejected fits of models that are no
    longer (or not yet) active -- against flow_pipeline.select_transitions; no libm anywhere, so the paths are bit-equal."""
    import torch

    from vstab_amd import flow_pipeline as fp
    from vstab_amd import native

    rng = np.random.default_rng(5)
    pairs = 41
    for case in range(7):
        table = np.zeros((pairs, 3), native.FIT_DTYPE)
        for i in range(pairs):
            for m in (0, 1, 2):
                mat = np.eye(3, dtype=np.float32)
                mat[0, 2], mat[1, 2] = rng.uniform(-3, 3, 2)
                if m >= 1:
                    c, s_ = np.cos(rng.uniform(-0.01, 0.01)), np.sin(rng.uniform(-0.01, 0.01))
                    mat[0, 0] = mat[1, 1] = 1.002 * c
                    mat[0, 1], mat[1, 0] = -1.002 * s_, 1.002 * s_
                if m == 2:
                    mat[2, 0], mat[2, 1] = rng.uniform(-2e-6, 2e-6, 2)
                    mat[0, 0] += rng.uniform(-1e-3, 1e-3)
                table[i, m]["matrix"] = mat.reshape(9)
                table[i, m]["computed"] = table[i, m]["accepted"] = 1
                table[i, m]["confidence"] = 0.9
        if case == 1:
            table[17, 2]["accepted"] = 0                                   # 2 -> 1 mid-clip
            table[30, 1]["accepted"] = 0                                   # 1 -> 0 later
        elif case == 2:
            table[0, 2]["accepted"] = 0                                    # at the first pair, straight down to translation
            table[0, 1]["computed"] = 0
        elif case == 3:
            table[pairs - 1, 2]["accepted"] = 0                            # at the last pair
        elif case == 4:
            table[8, 2]["accepted"] = table[8, 1]["accepted"] = table[8, 0]["accepted"] = 0   # nothing usable: identity, active = translation
            table[20, 0]["computed"] = 0                                   # ... and another identity pair later
        elif case == 5:
            table[3, 1]["accepted"] = 0                                    # lower models rejected while perspective still wins: ignored
            table[4, 0]["accepted"] = 0
            table[25, 2]["computed"] = 0                                   # 2 -> 1
            table[26, 2]["accepted"] = 1                                   # (a usable perspective fit after the step down is not taken up again)
        elif case == 6:
            table[10, 2]["accepted"] = 0
            table[10, 1]["accepted"] = 0                                   # 2 -> 0 directly
            table[11, 1]["accepted"] = 1
        dev = torch.from_numpy(table.view(np.uint8).reshape(-1).copy()).to(ctx.device)
        ctx.flow_plan_device(dev.data_ptr(), pairs, "perspective", (1920, 1080), (960, 540), 0.5, 16.0, 0.7, False)
        final_dev, path_dev, target_dev, _ = ctx.flow_plan_result(pairs + 1, 8)
        plan = fp.plan_stabilization(ctx, table, (1920, 1080), pairs + 1, "crop_and_pad", "perspective", False, 0.7, 0.5, 0.6,
                                     (127, 127, 127), 16.0, 16.0)
        assert np.array_equal(path_dev, plan.estimated_motion["path"]), case
        assert np.array_equal(target_dev, plan.estimated_motion["target_path"]), case
        # final = T @ M: sums of two inexact terms, whose float32 rounding is NumPy's matmul's business (fused or not): equal to a
        # few units in the last place here, bit-equal for nearly every frame (what is not is warped again by the pipeline)
        assert np.abs(final_dev.astype(np.float64) - plan.final_matrices).max() <= 1e-4 * 2.0 ** -20, case
        differing = int((final_dev.view(np.uint32) != plan.final_matrices.view(np.uint32)).reshape(pairs + 1, -1).any(axis=1).sum())
        assert differing <= 2, (case, differing)


@pytest.mark.parametrize("mode", ["similarity", "translation"])
def test_sticky_mode_on_the_device_equals_the_host_walk(ctx, pkg, mode):
    """The closed form of the sticky active_mode rule in plan_kernel (first unusable similarity fit -> translation from
    there on; no candidate -> identity) against flow_pipeline.select_transitions' pair-by-pair walk, on tables with
    rejected similarity fits, pairs without any candidate, and a rejected LAST / FIRST pair."""
    import torch

    from vstab_amd import flow_pipeline as fp
    from vstab_amd import native

    rng = np.random.default_rng(11)
    pairs = 37
    for case in range(6):
        table = np.zeros((pairs, 3), native.FIT_DTYPE)
        for i in range(pairs):
            for m in (0, 1):
                mat = np.eye(3, dtype=np.float32)
                mat[0, 2], mat[1, 2] = rng.uniform(-3, 3, 2)
                if m == 1:
                    c, s_ = np.cos(rng.uniform(-0.01, 0.01)), np.sin(rng.uniform(-0.01, 0.01))
                    mat[0, 0] = mat[1, 1] = 1.002 * c
                    mat[0, 1], mat[1, 0] = -1.002 * s_, 1.002 * s_
                table[i, m]["matrix"] = mat.reshape(9)
                table[i, m]["computed"] = 1 if (m == 0 or mode == "similarity") else 0
                table[i, m]["accepted"] = table[i, m]["computed"]
                table[i, m]["confidence"] = 0.9
        if case == 1:
            table[20, 1]["accepted"] = 0                                   # similarity rejected mid-clip: translation from there
        elif case == 2:
            table[0, 1]["accepted"] = 0                                    # ... at the first pair
        elif case == 3:
            table[pairs - 1, 1]["accepted"] = 0                            # ... at the last pair
        elif case == 4:
            table[9, 1]["accepted"] = 0
            table[9, 0]["accepted"] = 0                                    # no candidate at all: identity
            table[25, 0]["computed"] = 0
        elif case == 5:
            table[5, 0]["accepted"] = 0                                    # unusable translation while similarity is still active: ignored
            table[12, 1]["computed"] = 0
            table[30, 0]["accepted"] = 0
        dev = torch.from_numpy(table.view(np.uint8).reshape(-1).copy()).to(ctx.device)
        ctx.flow_plan_device(dev.data_ptr(), pairs, mode, (1920, 1080), (960, 540), 0.5, 16.0, 0.7, False)
        final_dev = ctx.flow_plan_result(pairs + 1, fp.PLAN_PARAMS[mode])[0]
        plan = fp.plan_stabilization(ctx, table, (1920, 1080), pairs + 1, "crop_and_pad", mode, False, 0.7, 0.5, 0.6,
                                     (127, 127, 127), 16.0, 16.0)
        # A wrong choice anywhere would move entries by ~1e-3.  Bit equality is NOT asserted here: after a fallback to
        # translation the rotation path is exactly constant, its smoothed value differs from it by a rounding error whose
        # presence depends on the last bit of the path (device atan2 vs glibc), and a correction of 0 vs 9e-19 rad is
        # visible in float32 -- the one situation in which the verification of the speculative plan does find differing
        # matrices and warps those frames again (test_a_wrong_device_plan_is_caught_and_the_frame_warped_again).
        assert np.abs(final_dev.astype(np.float64) - plan.final_matrices).max() < 1e-12, (mode, case)
        if case < 4:
            assert np.array_equal(final_dev.view(np.uint32), plan.final_matrices.view(np.uint32)), (mode, case)


def test_plan_kernel_reads_a_gathered_table_as_the_all_gather_leaves_it(ctx, pkg):
    """Multi-GPU layout: the receive buffer of the ranks' all-gather holds one block of `rows` records rows per rank of which
    the first pairs_of_rank are valid (rank 0 has one pair fewer than its frames; a rank without frames has none).  The plan
    formed from that buffer through the segment table equals the plan formed from the contiguous table, bit for bit."""
    import torch

    from vstab_amd import distributed as vd
    from vstab_amd import native

    frames = _clip(ctx, 23, 960, 540, "similarity", amp=1.5)
    pairs, work = _fits(ctx, frames, "similarity")
    table = ctx.sample_fit_batch_end(pairs)                      # [22, 3] records
    rec = 3 * native.FIT_DTYPE.itemsize
    raw = table.view(np.uint8).reshape(pairs, rec)
    args = ("similarity", (960, 540), work, 0.5, 16.0, 0.7, False)
    dev = torch.from_numpy(raw.reshape(-1).copy()).to(ctx.device)
    ctx.flow_plan_device(dev.data_ptr(), pairs, *args)
    want = ctx.flow_plan_result(pairs + 1, 4)
    for world in (2, 5, 8, 30):                                  # 30 ranks for 23 frames: ranks without frames
        per_rank = vd.transition_counts(pairs + 1, world)
        assert sum(per_rank) == pairs
        rows = max(max(per_rank), 1)
        flat = np.full((world, rows, rec), 0xAB, np.uint8)       # padding rows hold garbage: they must never be read
        at = 0
        for r, k in enumerate(per_rank):
            flat[r, :k] = raw[at:at + k]
            at += k
        d = torch.from_numpy(flat).to(ctx.device)
        ctx.flow_plan_device(d.data_ptr(), pairs, *args, seg_pairs=per_rank, seg_rows=rows)
        got = ctx.flow_plan_result(pairs + 1, 4)
        for a, b in zip(got, want):
            assert np.array_equal(a.view(np.uint64 if a.dtype == np.float64 else np.uint32), b.view(np.uint64 if b.dtype == np.float64 else np.uint32)), world
    with pytest.raises(native.VstabError, match="segments hold"):
        ctx.flow_plan_device(dev.data_ptr(), pairs, *args, seg_pairs=[3, 4], seg_rows=4)


def test_many_wrong_frames_are_rewarped_in_runs(ctx, pkg, monkeypatch):
    """_rewarp_mismatched with several runs of differing frames (a forged device-plan result: frames 2-4, 7 and 10-11 marked
    wrong): the runs are warped again in place, everything else is left alone, the count is returned."""
    import torch

    from vstab_amd import flow_pipeline as fp
    from vstab_amd import host_math as hm

    frames = _clip(ctx, 12, 640, 360, "similarity", amp=1.5)
    want, _ = _run(ctx, frames, "similarity", monkeypatch, device_plan=False)
    final = np.array([e["applied_matrix"] for e in want.meta["stabilization_warp"]["per_frame"]], np.float32)
    plan = fp.FlowPlan(final, (640, 360), {}, {}, {}, "crop_and_pad", (640, 360), 16.0)
    forged = final.copy()
    wrong = [2, 3, 4, 7, 10, 11]
    forged[wrong, 0, 2] += 1.0
    dst = want.frames.clone()
    mask = want.masks[..., 0].clone().contiguous()
    counts = torch.zeros(12, dtype=torch.int32, device=ctx.device)
    dst[wrong] = -1.0                                            # what a wrong plan would have left there
    n = fp._rewarp_mismatched(ctx, frames, plan, forged, dst, mask, counts, (127, 127, 127))
    assert n == 6 and bool((dst == want.frames).all()) and bool((mask == want.masks[..., 0]).all())
    untouched = [i for i in range(12) if i not in wrong]
    assert bool((counts[untouched] == 0).all())
