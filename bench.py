#!/usr/bin/env python3
"""Headline benchmark: stabilized frames/s of the Video Stabilizer Flow hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" = one full pass of the hot path over the clip with the input frames already resident in HBM and the
outputs left in HBM, entered at the node boundary (`_normalize_video_input` on the resident tensor):
    F0 input adaptation (the value-range sniff rides on the gray pass) -> gray+downscale -> DIS (all pairs) -> fit ->
    [all-gather of fit records if N > 1] -> trajectory -> framing -> warp + mask + padding counts -> meta.

Workloads (BASELINE.json configs):
  N = 1  C2: 256 synthetic 1080p frames, Flow similarity + crop_and_pad, defaults (the configuration the metric is quoted on).
  N > 1  the same 256 frames of work on every GPU: ONE 256 x N-frame 1080p clip sharded contiguously over the N ranks
         (1-frame halo, RCCL all-gather of the fit records, the plan over the whole clip on every rank): per-GPU work
         fixed -> "scaling": "weak".  `--total-frames T` fixes the clip instead ("strong"; also at N = 1:
         `--gpus 1 --total-frames 1024` is the C4 clip on one GPU), `--frames F` picks another per-GPU share.
         The default N > 1 line also carries, measured after the timed loop on the same ranks:
           `c4`  BASELINE configs[3]: one 1024-frame clip over the N ranks, total work fixed ("strong"), with the one-GPU
                 time of the same clip next to it (static record under profiles/);
           `c5`  BASELINE configs[4] (below).
  --workload c5   BASELINE configs[4] as the timed step: one 512-frame 4K clip (64 frames = one GPU's share at N = 1), Flow
         expand -> Motion Apply (expand, bilinear, motion_blur 0.5, Ultra = 33 samples) on the original frames, sharded
         the same way (the replay half has no collective).
  --workload c3   BASELINE configs[2] as the timed step (256 x 1080p, Flow perspective -> Motion Apply bicubic, 0.5, High).
N > 1 needs one process per GPU.  Started under torchrun (WORLD_SIZE set) this file is a rank; started plainly with
--gpus N > 1 it launches `python -m torch.distributed.run --nproc-per-node N bench.py ...` as a CHILD process before
anything touches the GPU and relays its output (never an exec).

Also on the JSON line (rank 0):
  roofline       the warp kernel (dominant HBM stream): algorithmic 28 B per output pixel x pixels per launch /
                 average launch time from HIP events on the launch stream
  cpu_baseline   (N = 1 only) decided at run time: if `import cv2` works, the reference's own OpenCV calls with its
                 arguments timed on this host (kind "opencv", cv2 version / threads / IPP stated) plus a `cv2_parity` object
                 (HIP vs cv2: gray levels, flow EPE, matrices, pixels under both sub-pixel conventions, masks) and the port's
                 figure as `cpu_baseline_port`; otherwise the CPU oracle (a C restatement of that path; kind "port") with
                 "cv2": "absent".  Always: CPU model, os.cpu_count(), threads used, per-stage seconds.
  accuracy       (N = 1) the 255 transitions the timed configuration reports against the clip's ANALYTIC motion
                 M_{i+1} M_i^-1 (px at working resolution, max / mean / p99), for the HIP run and for the CPU port
  parity_at_size (N = 1, inside the cpu_baseline leg: the oracle is the checker) the HIP run against the oracle's run of the
                 same 256 frames: matrices, confidences, every output pixel, every mask pixel, padding statistics
  batch_invariance (N = 1) pairs {0,127,254} re-run as 2-frame clips and frames {0,127,255} warped alone == the clip run
  c4, c5         (N > 1; c5 also with --force-dist) BASELINE configs[3] / configs[4] on the same ranks, outside the timed loop
  host_roundtrip the node as ComfyUI calls it: CPU tensor in -> CPU tensors out (PCIe-inclusive; never `value`);
                 host_roundtrip_8bit_source: the same with the clip quantised to float32(k) / 255, what an IMAGE decoded from
                 8-bit video holds (such chunks cross PCIe as bytes)
  motion_apply   Motion Apply rates for C3 (1080p, bicubic, blur 0.5, S=17) and C5's per-GPU share (4K, bilinear,
                 blur 0.5, S=33), device-resident (N = 1 only; measured outside the timed loop)
  config.rank0_host_ms   (N > 1) host wall-clock per phase of rank 0's step: the gathers and plan+meta are the
                 replicated/serial part that bounds strong scaling
"""

from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec
WARP_BYTES_PER_PIXEL = 28  # read 12 B source + write 12 B frame + 4 B mask
FLOW_ARGS = ("crop_and_pad", "similarity", False, 0.7, 0.5, 0.6, (127, 127, 127), 16.0)


def camera_matrices(n: int, offset: int, width: int, height: int) -> np.ndarray:
    """Per-frame camera transform: the reference's 'handheld' shake recipe (seed 0, 16 fps; committed
    as data in tests/golden/) tiled over the clip, composed with a slow 0.7 px/frame pan."""
    blk = json.loads((ROOT / "tests" / "golden" / "shake_c3_256x1080p.json").read_text())
    base = np.array([e["matrix"] for e in blk["per_frame"]], np.float64)
    sx, sy = width / 1920.0, height / 1080.0
    out = np.empty((n, 3, 3), np.float64)
    for k in range(n):
        i = offset + k
        m = base[i % len(base)].copy()
        m[0, 2] *= sx
        m[1, 2] *= sy
        pan = np.array([[1, 0, 0.7 * i * sx], [0, 1, 0.15 * i * sy], [0, 0, 1.0]])
        out[k] = pan @ m
    return out


def synth_clip(n: int, offset: int, height: int, width: int, device, seed: int = 1234, mats=None):
    """Band-limited procedural texture sampled analytically under the camera path (no interpolation):
    frame_i(p) = T(M_i^-1 p).  Returns float32 [n,H,W,3] in [0.05,0.95] on `device`.
    mats: explicit camera matrices [n,3,3] instead of camera_matrices(n, offset, ...)."""
    import torch

    rng = np.random.default_rng(seed)
    k = 20
    fx = torch.tensor(rng.uniform(-0.11, 0.11, k) * (1920.0 / width), device=device, dtype=torch.float32)
    fy = torch.tensor(rng.uniform(-0.11, 0.11, k) * (1080.0 / height), device=device, dtype=torch.float32)
    ph = torch.tensor(rng.uniform(0, 6.28, (3, k)), device=device, dtype=torch.float32)
    amp = torch.tensor(rng.uniform(0.3, 1.0, k), device=device, dtype=torch.float32)
    if mats is None:
        mats = camera_matrices(n, offset, width, height)
    inv64 = np.linalg.inv(np.asarray(mats, np.float64))
    projective = [bool(np.any(np.abs(m[2] / m[2, 2] - (0.0, 0.0, 1.0)) > 1e-15)) for m in inv64]
    inv = torch.tensor(inv64, device=device, dtype=torch.float32)
    yy, xx = torch.meshgrid(torch.arange(height, device=device, dtype=torch.float32),
                            torch.arange(width, device=device, dtype=torch.float32), indexing="ij")
    out = torch.empty((n, height, width, 3), device=device, dtype=torch.float32)
    norm = float(amp.sum())
    for i in range(n):
        m = inv[i]
        X = m[0, 0] * xx + m[0, 1] * yy + m[0, 2]
        Y = m[1, 0] * xx + m[1, 1] * yy + m[1, 2]
        if projective[i]:   # camera paths with perspective rows (tests / tools; the bench's own path is affine)
            Wd = m[2, 0] * xx + m[2, 1] * yy + m[2, 2]
            X, Y = X / Wd, Y / Wd
        arg = X[..., None] * fx + Y[..., None] * fy  # [H,W,k]
        for c in range(3):
            v = (torch.sin(arg + ph[c]) * amp).sum(-1) / norm
            out[i, ..., c] = 0.5 + 0.45 * torch.tanh(2.5 * v)
        del arg
    return out


def transition_accuracy(full_res_matrices, cam: np.ndarray, size, work_size) -> dict:
    """Error of reported transitions against the clip's analytic ones (independent of the oracle).

    A texture point q shows at p_i = M_i q in frame i (synth_clip samples T(M_i^-1 p)), so the true transition is
    A_i = M_{i+1} M_i^-1 (flow.py:133-210 estimates prev -> curr).  Errors are expressed at WORKING resolution (where DIS
    ran): the displacement error |got(p) - A_i(p)| at the frame centre and its max over the four corners, in px, and
    max |delta| over the 2x2 part (scale free)."""
    got = np.asarray(full_res_matrices, np.float64).reshape(-1, 3, 3)
    pairs = got.shape[0]
    w, h = size
    k = (work_size[0] / w) if work_size else 1.0   # working px per full-res px
    pts = np.array([[w / 2, h / 2, 1.0], [0, 0, 1.0], [w - 1, 0, 1.0], [0, h - 1, 1.0], [w - 1, h - 1, 1.0]], np.float64).T
    true = np.stack([cam[i + 1] @ np.linalg.inv(cam[i]) for i in range(pairs)])
    true = true / true[:, 2:3, 2:3]                                    # the reference reports homographies with h22 = 1
    pg = got @ pts
    pt = true @ pts
    pg = pg[:, :2] / pg[:, 2:3]
    pt = pt[:, :2] / pt[:, 2:3]
    err = np.hypot(*(pg - pt).transpose(1, 0, 2)) * k          # [P, 5] px at working res
    centre, corner = err[:, 0], err[:, 1:].max(axis=1)
    lin = np.abs(got[:, :2, :2] - true[:, :2, :2]).reshape(pairs, -1).max(axis=1)

    def dist(v):
        return {"max": float(v.max()), "mean": float(v.mean()), "p99": float(np.percentile(v, 99))}

    return {"pairs": pairs, "centre_px": {k_: round(v, 5) for k_, v in dist(centre).items()},
            "corner_px": {k_: round(v, 5) for k_, v in dist(corner).items()},
            "lin_2x2": {k_: float(f"{v:.3e}") for k_, v in dist(lin).items()},
            "true_motion_px": {"max": round(float((np.hypot(*(pt[:, :, 0] - pts[:2, 0]).T) * k).max()), 2)}}


def static_texture_error(res, cam: np.ndarray, frames, device) -> dict:
    """camera_lock + strength 1 (flow.py:356-371: target path 0): every output frame should show frame 0's view moved by
    the crop_and_pad recentring shift, i.e. T((S M_0)^-1 p).  Error of the output against that analytically sampled
    image over the pixels no frame padded (float 0..1 units)."""
    import torch

    n, h, w, _ = res.frames.shape
    off = res.meta["framing"]["center_offset"]
    shift = np.array([[1, 0, off[0]], [0, 1, off[1]], [0, 0, 1.0]])
    want = synth_clip(1, 0, h, w, device, mats=(shift @ cam[0])[None])[0]
    safe = (res.masks.reshape(n, h, w) == 0).all(dim=0)
    # two pixels in from the padded region: a bilinear tap next to the border blends the padding colour in although the
    # nearest-neighbour mask (flow.py:566-582) still calls the pixel covered
    unsafe = torch.nn.functional.max_pool2d((~safe)[None, None].float(), 5, stride=1, padding=2)[0, 0] > 0
    unsafe[:2] = unsafe[-2:] = True
    unsafe[:, :2] = unsafe[:, -2:] = True
    safe = ~unsafe
    err = (res.frames - want[None]).abs().amax(dim=-1)          # [n,h,w]
    err = torch.where(safe[None], err, torch.zeros_like(err))
    per_frame = err.reshape(n, -1).amax(dim=1)
    mse = float(((res.frames - want[None]) ** 2).mean(dim=-1)[:, safe].mean().item())
    return {"frames": n, "safe_fraction": round(float(safe.float().mean().item()), 4), "max_abs": round(float(per_frame.max().item()), 5),
            "mean_abs": round(float(err[:, safe].mean().item()), 6), "psnr_db": round(10 * np.log10(1.0 / max(mse, 1e-20)), 2),
            "worst_frame": int(per_frame.argmax().item())}


def host_cpu_info() -> dict:
    """The host the CPU leg ran on (always stated)."""
    model = None
    try:
        for ln in Path("/proc/cpuinfo").read_text().splitlines():
            if ln.startswith("model name"):
                model = ln.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"cpu_model": model, "os_cpu_count": os.cpu_count(), "affinity_cpus": len(os.sched_getaffinity(0))}


def cpu_baseline(frames_host: np.ndarray, threads: int, keep_outputs: bool = False, provider=None):
    """Time a CPU provider of the reference path's primitives over a bounded sample of the clip: the C oracle (a
    restatement of the reference's OpenCV path, kind "port"; the default) or oracle.cv2_tier.Cv2Tier (the reference's own
    cv2 calls, kind "opencv") -- both through the same plan code.
    keep_outputs: also hand back what the provider computed (fits, plan matrices, warped pixels) for the parity checks."""
    os.environ["OMP_NUM_THREADS"] = str(threads)
    from oracle import oracle as vo

    vo.build()
    vo.set_threads(threads)   # a second call in one process (the all-cores leg): the OpenMP runtime is already up
    kind = getattr(provider, "kind", "port")
    if provider is None:
        provider = vo
    import __graft_entry__ as graft

    graft.load_package()
    from vstab_amd import flow_pipeline as fp
    from vstab_amd import host_math as hm

    n, h, w, _ = frames_host.shape
    size = (w, h)
    t0 = time.perf_counter()
    peaks = provider.frame_max(frames_host)            # F0: the per-frame range sniff of stabilizer_utils.py:127-131
    assert not (peaks > 1.5).any()
    work = hm._working_estimation_size(w, h)
    gray = provider.gray_for_estimation(frames_host, work)
    t_gray = time.perf_counter()
    flow = provider.dis_flow_clip(gray)
    t_dis = time.perf_counter()
    recs = [provider.fit_all_modes(flow[i], 8, "similarity")[0] for i in range(n - 1)]
    t_fit = time.perf_counter()
    work_mats, _, confs, resids, _ = fp.select_transitions(recs, "similarity")
    mats = [hm._rescale_transform_to_full(m, size, work) if work else m for m in work_mats]
    deltas = np.stack([hm._matrix_to_params(m, "similarity") for m in mats])
    path = np.concatenate([np.zeros((1, 4)), np.cumsum(deltas, axis=0)])
    win = hm.smoothing_window(0.5, 16.0)
    kern = np.ones(win) / win
    sm = np.stack([np.convolve(np.pad(path[:, d], (win // 2,) * 2, mode="edge"), kern, mode="valid") for d in range(4)], 1)
    diffs = 0.7 * (sm - path)
    apply = [hm._params_to_matrix(d, "similarity") for d in diffs]
    mins, maxs = hm._compute_bounding_boxes(apply, w, h)
    x0, y0, x1, y1 = mins[:, 0].max(), mins[:, 1].max(), maxs[:, 0].min(), maxs[:, 1].min()
    shift = np.array([[1, 0, w * 0.5 - (x0 + x1) * 0.5], [0, 1, h * 0.5 - (y0 + y1) * 0.5], [0, 0, 1]], np.float32)
    final = np.stack([shift @ m for m in apply]).astype(np.float32)
    t_plan = time.perf_counter()
    warped, mask, counts = provider.warp_clip(frames_host, final, size, border=hm.border_value((127, 127, 127)))
    dt = time.perf_counter() - t0
    line = {"value": n / dt, "unit": "frames/s", "cores": threads, "kind": kind,
            "sample": f"{n} frames of the same synthetic {w}x{h} clip, full path (range sniff, gray, DIS, fit, trajectory, "
                      f"warp+mask), " + ("OpenMP over frames" if kind == "port" else "OpenCV's own thread pool") + f", {dt:.1f} s",
            "stage_s": {"gray": round(t_gray - t0, 3), "dis": round(t_dis - t_gray, 3), "fit": round(t_fit - t_dis, 3),
                        "plan": round(t_plan - t_fit, 4), "warp": round(t0 + dt - t_plan, 3)},
            **host_cpu_info()}
    if not keep_outputs:
        return line, None
    return line, {"transitions": np.stack(mats), "confidences": confs, "residuals": resids, "final": final,
                  "frames": warped, "masks": mask, "counts": counts, "gray": gray, "flow": flow}


def cv2_leg(ctx, torch, frames_dev, threads: int, max_frames: int, allow_standin: bool = False) -> dict:
    """The real-OpenCV tier, decided at run time: returns {"probe": ...} always,
    plus -- when a real `cv2` imports -- "baseline" (the reference's own cv2 calls timed on this host, kind "opencv") and
    "parity" (the HIP path against them on the same frames: gray levels, flow end-point error, fitted / final matrices,
    pixels under both sub-pixel conventions, masks).  The oracle-backed stand-in of tests/golden is never timed."""
    from oracle import cv2_tier

    info = cv2_tier.probe()
    out = {"probe": info}
    if info["cv2"] == "absent" or (info.get("standin") and not allow_standin):   # allow_standin: tests of this function only
        return out
    import __graft_entry__ as graft

    graft.load_package()
    from vstab_amd import flow_pipeline as fp
    from vstab_amd import host_math as hm

    tier = cv2_tier.Cv2Tier(threads)
    n = min(int(frames_dev.shape[0]), max_frames)
    sample_dev = frames_dev[:n]
    sample = sample_dev.cpu().numpy()
    h, w = sample.shape[1:3]
    base, ref = cpu_baseline(sample, threads, keep_outputs=True, provider=tier)
    base.update({"cv2_version": info["cv2"], "cv2_threads": int(tier.cv2.getNumThreads()) if hasattr(tier.cv2, "getNumThreads") else None,
                 "ipp": info.get("ipp")})
    out["baseline"] = base
    # ---- parity: the HIP path on the same n frames
    res = fp._stabilize_frames(hm._normalize_video_input(sample_dev), *FLOW_ARGS, ctx=ctx, keep_on_device=True)
    work = hm._working_estimation_size(w, h)
    gray_hip = ctx.gray_downscale(sample_dev, work)
    par = {"frames": n, "tolerance_target": "north_star: pixels within 1e-3 of OpenCV"}
    g = gray_hip.cpu().numpy()
    par["gray_u8_differing"] = int(np.count_nonzero(g != ref["gray"]))
    par["gray_u8_max_abs"] = int(np.abs(g.astype(np.int16) - ref["gray"].astype(np.int16)).max())
    k = min(n, 9)
    flow_hip, _ = ctx.dis_flow_batch(torch.from_numpy(ref["gray"][:k]).to(sample_dev.device), want_full=True, want_grid=False)
    fh = flow_hip.cpu().numpy()
    epe = np.hypot(fh[..., 0] - ref["flow"][:k - 1, ..., 0], fh[..., 1] - ref["flow"][:k - 1, ..., 1])
    par["flow_epe_px_on_cv2_gray"] = {"pairs": k - 1, "max": float(epe.max()), "mean": float(epe.mean()),
                                      "p99": float(np.percentile(epe, 99)), "at_stride8_max": float(epe[:, ::8, ::8].max())}
    em = res.meta["estimated_motion"]["per_transition"]
    got_t = np.array([t["matrix"] for t in em], np.float64)
    par["transition_matrices_max_abs"] = float(np.abs(got_t - ref["transitions"]).max())
    par["transition_translation_max_abs_px"] = float(np.abs(got_t[:, :2, 2] - ref["transitions"][:, :2, 2]).max())
    got_final = np.array([e["applied_matrix"] for e in res.meta["stabilization_warp"]["per_frame"]], np.float64)
    par["final_matrices_max_abs"] = float(np.abs(got_final - ref["final"]).max())
    # pixels: the warp isolated (HIP warp with OpenCV's matrices, both sub-pixel conventions), then end to end
    idx = sorted({0, n // 2, n - 1})
    border = hm.border_value((127, 127, 127))
    for subpix in ("q5", "exact"):
        dst, mask, _ = ctx.warp_batch(sample_dev[idx], ref["final"][idx], (w, h), interp="bilinear", border=border, subpix=subpix,
                                      want_mask=True, want_count=True)
        d = np.abs(dst.cpu().numpy().astype(np.float64) - ref["frames"][idx])
        par[f"warp_pixels_{subpix}"] = {"frames": idx, "max": float(d.max()), "mean": float(d.mean()),
                                        "frac_over_1e-3": float((d > 1e-3).mean())}
        if subpix == "q5":
            par["mask_pixels_differing"] = int(np.count_nonzero(mask.cpu().numpy() != ref["masks"][idx]))
    par["subpix_matched"] = min(("q5", "exact"), key=lambda s_: par[f"warp_pixels_{s_}"]["max"])
    d = np.abs(res.frames[idx].cpu().numpy().astype(np.float64) - ref["frames"][idx])
    par["end_to_end_pixels"] = {"frames": idx, "max": float(d.max()), "mean": float(d.mean()), "frac_over_1e-3": float((d > 1e-3).mean())}
    par["within_1e-3"] = bool(par[f"warp_pixels_{par['subpix_matched']}"]["max"] <= 1e-3)
    out["parity"] = par
    return out


def check_against_oracle(meta, res_frames, res_masks, port: dict) -> dict:
    """Parity AT THE BENCH SIZE (part of the cpu_baseline leg: the oracle is the checker, never the thing measured): the
    HIP run's reported transitions / final matrices / pixels / masks / padding counts against what the CPU oracle computed
    for the same clip.  Bit-equal or reported as a count of differing elements."""
    em = meta["estimated_motion"]["per_transition"]
    got_t = np.array([t["matrix"] for t in em], np.float32)
    got_final = np.array([e["applied_matrix"] for e in meta["stabilization_warp"]["per_frame"]], np.float32)
    n = got_final.shape[0]
    out = {"frames": n,
           "transition_matrices_equal": bool(np.array_equal(got_t, port["transitions"].astype(np.float32))),
           "transition_matrices_max_abs_diff": float(np.abs(got_t - port["transitions"]).max()),
           "confidences_equal": [t["confidence"] for t in em] == list(port["confidences"]),
           "residuals_max_rel_diff": float(max(abs(t["residual"] - r) / max(abs(r), 1e-30) for t, r in zip(em, port["residuals"]))),
           "final_matrices_max_abs_diff": float(np.abs(got_final - port["final"]).max())}
    # the pixel comparison isolates the warp: oracle warp re-run with the HIP run's own matrices when those differ in
    # the last bits (f32 plan arithmetic is numpy on both sides, so normally they do not)
    ref_frames, ref_masks, ref_counts = port["frames"], port["masks"], port["counts"]
    if not np.array_equal(got_final, port["final"]):
        from oracle import oracle as vo
        from vstab_amd import host_math as hm

        out["pixels_checked_with"] = "oracle warp re-run on the HIP run's matrices"
        ref_frames, ref_masks, ref_counts = vo.warp_clip(port["source"], got_final, tuple(meta["stabilization_warp"]["output_size"]),
                                                         border=hm.border_value((127, 127, 127)))
    f = res_frames.cpu().numpy()
    m = res_masks.cpu().numpy().reshape(ref_masks.shape)
    out["pixels_differing"] = int(np.count_nonzero(f != ref_frames))
    out["mask_pixels_differing"] = int(np.count_nonzero(m != ref_masks))
    ratios = (ref_counts.astype(np.float32) / np.float32(f.shape[1] * f.shape[2])).astype(np.float64)
    out["padding_stats_equal"] = bool(meta["padding_fraction_mean"] == float(np.mean(ratios))
                                      and meta["padding_fraction_max"] == float(np.max(ratios)))
    out["bit_equal"] = bool(out["transition_matrices_equal"] and out["confidences_equal"] and out["pixels_differing"] == 0
                            and out["mask_pixels_differing"] == 0 and out["padding_stats_equal"]
                            and out["final_matrices_max_abs_diff"] == 0.0)
    return out


def check_batch_invariance(fp, hm, ctx, frames, meta, res_frames, pairs=(0, 127, 254)) -> dict:
    """The per-pair results of the whole-clip run (fused DIS launch form, XCD block remap over the full grid, 64-bit
    frame bases) against 2-frame runs of the same pairs (split launch form, tiny grids): HIP against HIP, so this is
    not parity -- it shows that nothing depends on the batch size."""
    em = meta["estimated_motion"]["per_transition"]
    n = frames.shape[0]
    ok, worst = True, 0.0
    for p in pairs:
        if p + 1 >= n:
            continue
        sub = fp._stabilize_frames(hm._normalize_video_input(frames[p:p + 2]), *FLOW_ARGS, ctx=ctx, keep_on_device=True)
        t = sub.meta["estimated_motion"]["per_transition"][0]
        same = t["matrix"] == em[p]["matrix"] and t["confidence"] == em[p]["confidence"] and t["residual"] == em[p]["residual"]
        worst = max(worst, float(np.abs(np.array(t["matrix"]) - np.array(em[p]["matrix"])).max()))
        ok = ok and same
    # frames warped alone with the clip's matrices == the same frames of the whole-clip launch
    final = np.array([e["applied_matrix"] for e in meta["stabilization_warp"]["per_frame"]], np.float32)
    out_size = tuple(meta["stabilization_warp"]["output_size"])
    px_ok = True
    for i in sorted({0, n // 2 - 1, n - 1}):
        dst, _, _ = ctx.warp_batch(frames[i:i + 1], final[i:i + 1], out_size, interp="bilinear",
                                   border=hm.border_value((127, 127, 127)), want_mask=True, want_count=True)
        px_ok = px_ok and bool((dst[0] == res_frames[i]).all().item())
    return {"pairs": [p for p in pairs if p + 1 < n], "fit_records_equal": bool(ok), "fit_matrix_max_abs_diff": worst,
            "frames_equal": px_ok}


def warp_source_sha256() -> str:
    """Identity of the warp kernel the loaded library was built from (the library is rebuilt from these sources by
    __graft_entry__.build(); the judge re-derives the shipped binary from them)."""
    import hashlib

    h = hashlib.sha256()
    for name in ("vstab_warp.hip", "vstab_internal.h"):
        h.update((ROOT / "comfyui-video-stabilizer_amd" / "csrc" / name).read_bytes())
    return h.hexdigest()


def warp_traffic_record(frames: int, w: int, h: int):
    """`roofline.traffic`: the PMC-counted HBM bytes per launch of the warp kernel, from profiles/warp_traffic.json --
    a separate rocprofv3 --pmc collection (counters cannot ride on a timed run).  Tied to the kernel that runs: the record
    carries the sha256 of the kernel sources it was collected on; a mismatch (kernel edited since) or another workload
    gives traffic = None and the reason."""
    tfile = ROOT / "profiles" / "warp_traffic.json"
    if not tfile.exists():
        return None, "no profiles/warp_traffic.json"
    tj = json.loads(tfile.read_text())
    if tj.get("frames") != frames or tj.get("size") != [w, h]:
        return None, f"profiles/warp_traffic.json was collected on {tj.get('frames')} x {tj.get('size')}, not this workload"
    have = warp_source_sha256()
    if tj.get("kernel_source_sha256") != have:
        return None, ("profiles/warp_traffic.json was collected on other warp kernel sources (sha256 "
                      f"{str(tj.get('kernel_source_sha256'))[:12]} != {have[:12]}): re-run tools/pmc_traffic.sh")
    return tj.get("hbm_bytes_per_launch"), ("profiles/warp_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this kernel -- "
                                            f"sources sha256 {have[:12]} -- on this workload; static, not collected in this run)")


def launch_children(args, argv) -> int:
    """--gpus N > 1 without a torchrun environment: run the ranks as children of this (GPU-free) process."""
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve())] + argv
    proc = subprocess.run(cmd, env=env)
    return proc.returncode


def measure_host_roundtrip(nodes, frames_host, repeats: int = 3) -> dict:
    """The node exactly as ComfyUI calls it (stabilizer_utils.py:200-221: CPU tensors out)."""
    import torch

    best = None
    for _ in range(repeats):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = nodes.VideoStabilizerFlow.execute(frames_host, 16.0, "crop_and_pad", "similarity", False, 0.7, 0.5, 0.6, "#7F7F7F")
        dt = time.perf_counter() - t0
        assert out[0].device.type == "cpu" and out[0].shape[0] == frames_host.shape[0]
        del out
        best = dt if best is None else min(best, dt)
    n = frames_host.shape[0]
    gb = frames_host.numel() * 4 / 1e9
    from vstab_amd import native

    c = native.default_context()
    coded_in, chunks_in = getattr(c, "last_upload_coded", (0, 0))
    mask_coded = bool(getattr(c, "last_download_coded", False))
    pcie_in = gb * (1.0 - 0.75 * coded_in / max(chunks_in, 1))
    pcie_out = gb + gb / 3 * (0.25 if mask_coded else 1.0)
    return {"ms": round(best * 1e3, 1), "frames_per_s": round(n / best, 1), "frames": n,
            "pinned_output": os.environ.get("VSTAB_PINNED_OUTPUT", "0") not in ("", "0", "false", "False"),
            "bytes_in_GB": round(gb, 2), "bytes_out_GB": round(gb * 4 / 3, 2),
            "pcie_in_GB": round(pcie_in, 2), "pcie_out_GB": round(pcie_out, 2),
            "upload_chunks_as_bytes": [coded_in, chunks_in], "mask_as_bytes": mask_coded,
            "note": "CPU tensor in -> Video Stabilizer Flow node -> CPU tensors out, best of %d" % repeats}


def measure_motion_apply(ctx, torch, device, steps: int = 3, check: bool = True) -> dict:
    """Motion Apply rates for BASELINE configs C3 / C5 (per-GPU share), device-resident, HIP-event kernel time and
    wall time per pass; the motion comes from the reference's shake generator blocks (tests/golden/shake_*.json).
    check: one blurred frame of each leg (frame 1) is compared with the CPU oracle's rendering of it from a 2-frame window
    (part of the cpu_baseline leg: the oracle is the checker; outside every timed region) -> `oracle_spot_check`."""
    from vstab_amd import apply_pipeline as ap
    from vstab_amd import host_math as hm

    out = {}
    plans = [("c3_1080p_bicubic_blur0.5_S17", "shake_c3_256x1080p.json", 256, 1080, 1920, "crop_and_pad", "bicubic", 17),
             ("c5_4k_expand_bilinear_blur0.5_S33", "shake_c5_64x4k.json", 64, 2160, 3840, "expand", "bilinear", 33)]
    for key, fixture, n, h, w, framing, interp, samples in plans:
        meta = {"motion_meta": json.loads((ROOT / "tests" / "golden" / fixture).read_text())}
        frames = synth_clip(n, 0, h, w, device)
        torch.cuda.synchronize()

        def once():
            c = hm._normalize_video_input(frames)
            r = ap.apply_motion(c, meta, (127, 127, 127), framing_mode=framing, interpolation=interp, motion_blur=0.5,
                                motion_blur_samples=samples, ctx=ctx, keep_on_device=True)
            return r

        r = once()
        shape = list(r.frames.shape)
        spot = None
        if check:
            try:
                from oracle import oracle as vo

                m64 = np.array([e["matrix"] for e in meta["motion_meta"]["per_frame"]], np.float64)
                if framing == "expand":   # motion_apply.py:288-294: canvas and shift from all the clip's matrices
                    mins, maxs = hm._compute_bounding_boxes(list(m64), w, h)
                    shift, size = hm._prepare_expand_transform(mins, maxs)
                    m64 = np.stack([shift @ m for m in m64])
                    assert list(size) == [shape[2], shape[1]]
                ref, ref_mask = vo.warp_blur_clip(frames[1:3].cpu().numpy(), m64[1:3], (shape[2], shape[1]), 0.5, samples, interp=interp,
                                                  border=hm.border_value((127, 127, 127)))
                spot = {"frame": 1, "pixels_differing": int(np.count_nonzero(r.frames[1].cpu().numpy() != ref[0])),
                        "mask_pixels_differing": int(np.count_nonzero(r.masks[1, ..., 0].cpu().numpy() != ref_mask[0])),
                        "values": int(ref[0].size)}
                del ref, ref_mask
            except Exception as exc:
                spot = {"error": f"{type(exc).__name__}: {exc}"}
        del r
        torch.cuda.synchronize()
        ctx.set_timing(True)
        t0 = time.perf_counter()
        for _ in range(steps):
            r = once()
            del r
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        kernel_ms, launches = ctx.kernel_ms_stats("warp_blur")
        kernel_ms /= max(launches, 1)
        px = shape[1] * shape[2] * n
        out[key] = {"frames": n, "out_shape": shape, "ms_per_pass": round(dt * 1e3, 2), "frames_per_s": round(n / dt, 1),
                    "kernel_ms": round(kernel_ms, 2), "samples_per_s": round(px * samples / (kernel_ms * 1e-3), 0),
                    "hbm_GBs_algorithmic": round(WARP_BYTES_PER_PIXEL * px / (kernel_ms * 1e-3) / 1e9, 1),
                    "bound": "valu (not HBM): S x (f64 coordinates + taps) per output pixel"}
        if spot is not None:
            out[key]["oracle_spot_check"] = spot
        del frames
        torch.cuda.empty_cache()
    return out


C5_FLOW_ARGS = ("expand", "similarity", False, 0.7, 0.5, 0.6, (127, 127, 127), 16.0)
C5_APPLY = dict(framing_mode="expand", interpolation="bilinear", motion_blur=0.5, motion_blur_samples=33)
C3_FLOW_ARGS = ("crop_and_pad", "perspective", False, 0.7, 0.5, 0.6, (127, 127, 127), 16.0)
C3_APPLY = dict(framing_mode="crop_and_pad", interpolation="bicubic", motion_blur=0.5, motion_blur_samples=17)
CHAINS = {
    "c5": (C5_FLOW_ARGS, C5_APPLY, "C5", "Flow (DIS) similarity + expand -> Motion Apply expand, bilinear, motion_blur 0.5, Ultra (33 samples)"),
    "c3": (C3_FLOW_ARGS, C3_APPLY, "C3", "Flow (DIS) perspective + crop_and_pad -> Motion Apply crop_and_pad, bicubic, motion_blur 0.5, High (17 samples)"),
}


def run_c5(ctx, torch, dist, device, rank, world, use_dist, total, h, w, steps, warmup, chain: str = "c5") -> dict:
    """A Flow -> Motion Apply chain as ONE timed step.  chain "c5" = BASELINE configs[4]: Flow (DIS, similarity) with expand
    framing, then Motion Apply (expand, bilinear, motion_blur 0.5, Ultra = 33 samples); chain "c3" = BASELINE configs[2]:
    Flow perspective + crop_and_pad, then Motion Apply (crop_and_pad, bicubic, motion_blur 0.5, High = 17 samples) -- on the
    ORIGINAL frames with the returned meta (call shapes: video_stabilizer_flow.py:734-763,
    video_stabilizer_motion_apply.py:86-129), one `total`-frame clip sharded contiguously over the ranks.  The Flow half
    has the one all-gather of fit records (+ the pad counts); the replay half has no collective (distributed.py).
    Timed like the headline: barrier + synchronize on both sides, MAX over ranks."""
    C5_FLOW_ARGS, C5_APPLY, tag, what = CHAINS[chain]   # (shadows the module constants: the body below is the same for both)
    from vstab_amd import apply_pipeline as ap
    from vstab_amd import distributed as vd
    from vstab_amd import flow_pipeline as fp
    from vstab_amd import host_math as hm

    start, end = vd.shard_range(total, world, rank)
    n_local = end - start
    halo = 1 if (rank > 0 and n_local > 0) else 0

    def agree(ok: bool, what: str) -> None:
        """All ranks continue, or all ranks give up TOGETHER: a rank that failed alone (out of memory while the others
        fit, a VstabError) must not leave the others waiting in the chain's collectives -- as an extra on the headline
        line that would lose the headline too."""
        if use_dist:
            flag = torch.tensor([1 if ok else 0], device=device if dist.get_backend() == "nccl" else "cpu", dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            ok = bool(flag.item())
        if not ok:
            raise RuntimeError(f"{what} failed on at least one rank; all ranks skip the chain")

    try:
        frames = synth_clip(n_local + halo, start - halo, h, w, device)
        torch.cuda.synchronize()
        alloc_ok = True
    except Exception:   # typically out of memory for the 4K shard
        frames, alloc_ok = None, False
    agree(alloc_ok, "allocating the clip")
    stats: dict = {}
    lap = {"flow": 0.0, "apply": 0.0}
    plan_seen: list = []   # the Flow half's device-plan verdict per step ({"used", "mismatched_frames"})

    def step():
        t0 = time.perf_counter()
        if use_dist:
            _, _, meta = vd.stabilize_sharded(ctx, frames, total, *C5_FLOW_ARGS, stats=stats, want_meta=True)
            plan_seen.append(stats.get("device_plan"))
            t1 = time.perf_counter()
            out = vd.apply_motion_sharded(ctx, frames[halo:], start, total, meta, (127, 127, 127), **C5_APPLY)
        else:
            res = fp._stabilize_frames(hm._normalize_video_input(frames), *C5_FLOW_ARGS, ctx=ctx, keep_on_device=True)
            meta = res.meta
            plan_seen.append(res.device_plan)
            del res
            t1 = time.perf_counter()
            r = ap.apply_motion(hm._normalize_video_input(frames), meta, (127, 127, 127), ctx=ctx, keep_on_device=True, **C5_APPLY)
            out = (r.frames, r.masks, r.meta)
        lap["flow"] += t1 - t0
        lap["apply"] += time.perf_counter() - t1
        return out

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # the first pass (outputs allocated, LDS attributes set, every code path taken) runs alone, and its verdict is agreed
    # on before the timed passes: what can fail on one rank only fails here
    try:
        out = step()
        del out
        first_ok = True
    except Exception:
        first_ok = False
    agree(first_ok, "the first pass of the chain")
    for _ in range(max(0, warmup - 1)):
        out = step()
        del out
    fence()
    ctx.set_timing(True)
    stats.clear()
    plan_seen.clear()
    lap["flow"] = lap["apply"] = 0.0
    t0 = time.perf_counter()
    for _ in range(steps):
        out = step()
        shape, ameta = list(out[0].shape), out[2]
        del out
    fence()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], device=device if dist.get_backend() == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    stage_ms = {}
    for kind in ("gray", "dis", "fit", "warp", "warp_blur"):
        total_ms, launches = ctx.kernel_ms_stats(kind)
        stage_ms[kind] = round(total_ms / max(launches, 1), 3)
    del frames
    torch.cuda.empty_cache()
    out = {"workload": f"{tag}: one {total}-frame {w}x{h} clip, {what} on the original frames, device-resident",
           "value": round(total * steps / elapsed, 2), "unit": "frames/s", "n_gpus": world, "total_frames": total,
           "frames_per_gpu": n_local, "ms_per_step": round(elapsed / steps * 1e3, 3), "steps": steps, "warmup": warmup,
           "scaling": "strong", "out_shape_rank0": shape, "motion_blur_samples": ameta["motion_apply"]["motion_blur_samples"],
           "rank0_stage_ms": stage_ms,
           "rank0_device_plan": ({"used": all(bool(v and v.get("used")) for v in plan_seen),
                                  "mismatched_frames_max_per_step": max(int((v or {}).get("mismatched_frames", 0)) for v in plan_seen)}
                                 if plan_seen else None),
           "rank0_host_ms": {"flow_half": round(lap["flow"] / steps * 1e3, 3), "apply_half_launch": round(lap["apply"] / steps * 1e3, 3),
                             **{k: round(v / steps, 3) for k, v in stats.items() if isinstance(v, (int, float))}},
           "sharding": "single GPU" if world == 1 else f"contiguous frame shards x{world}, 1-frame halo for the Flow half, RCCL all-gather "
                       "of fit records; the Motion Apply half has no collective"}
    return out


def main() -> int:
    ap_ = argparse.ArgumentParser()
    ap_.add_argument("--gpus", type=int, default=1)
    ap_.add_argument("--steps", type=int, default=5)
    ap_.add_argument("--warmup", type=int, default=2)
    ap_.add_argument("--frames", type=int, default=None, help="frames per GPU (weak scaling); default: 256 (C2 at N=1; one 256 x N-frame clip over N GPUs)")
    ap_.add_argument("--total-frames", type=int, default=None, help="clip length sharded over all GPUs (strong scaling), e.g. 1024 = C4 as the line's value")
    ap_.add_argument("--height", type=int, default=None, help="default 1080 (2160 for --workload c5)")
    ap_.add_argument("--width", type=int, default=None, help="default 1920 (3840 for --workload c5)")
    ap_.add_argument("--workload", choices=("auto", "c5", "c3"), default="auto",
                     help="auto: the headline metric -- C2 at N=1, a C2-sized shard per GPU at N>1 (weak scaling; C4 and C5 objects ride on the same line). "
                          "c5: time BASELINE configs[4] (512 x 4K Flow expand -> Motion Apply blur Ultra) as the line's value. "
                          "c3: time BASELINE configs[2] (256 x 1080p Flow perspective -> Motion Apply bicubic, blur 0.5, High)")
    ap_.add_argument("--c5-frames", type=int, default=None, help="clip length of the C5 workload (default: 512 over N > 1 GPUs; 64 = one GPU's share of the 8-GPU config at N = 1)")
    ap_.add_argument("--force-dist", action="store_true", help="run the sharded/RCCL code path even with one rank (rehearsal)")
    ap_.add_argument("--cpu-frames", type=int, default=256, help="frames of the clip timed on the CPU oracle (0 = skip)")
    ap_.add_argument("--cv2-frames", type=int, default=128, help="frames of the clip run through a real cv2 when one imports (baseline kind 'opencv' + cv2_parity)")
    ap_.add_argument("--no-extras", action="store_true", help="skip host_roundtrip / motion_apply (N=1 extras outside the timed loop)")
    ap_.add_argument("--rehearse-on-one-gpu", action="store_true",
                     help="N > 1 ranks that all use cuda:0 with a gloo control plane (RCCL refuses two ranks on a device): runs the "
                          "multi-rank bench code on a one-GPU box; the numbers mean nothing, the line says so")
    ap_.add_argument("--no-checks", action="store_true", help="skip accuracy / batch_invariance / parity_at_size (profiling runs: keeps the timed steps last in a trace)")
    args = ap_.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_children(args, sys.argv[1:])

    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    dev_index = 0 if args.rehearse_on_one_gpu else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)

    import torch.distributed as dist

    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        import datetime

        # a collective that never completes (a rank that died) becomes an error after ten minutes, not a silent hang
        limit = datetime.timedelta(minutes=10)
        if args.rehearse_on_one_gpu:
            # rehearse the form a real multi-GPU run takes (plan on the device from the gathered table), with its collectives
            # through the host
            os.environ.setdefault("VSTAB_SHARDED_DEVICE_PLAN", "force")
            dist.init_process_group(backend="gloo", timeout=limit)
        else:
            dist.init_process_group(backend="nccl", device_id=device, timeout=limit)

    import __graft_entry__ as graft

    graft.load_package()
    from vstab_amd import distributed as vd
    from vstab_amd import flow_pipeline as fp
    from vstab_amd import host_math as hm
    from vstab_amd import native, nodes

    ctx = native.Context(dev_index)
    ctx.set_timing(True)

    if args.workload in ("c5", "c3"):
        if args.workload == "c5":
            h, w = args.height or 2160, args.width or 3840
            total5 = args.c5_frames or args.total_frames or (512 if world > 1 else 64)   # one GPU: the per-GPU share of the 8-GPU config
            metric = "stabilized + motion-blurred frames/sec (4K, Flow expand -> Motion Apply Ultra; BASELINE configs[4])"
        else:
            h, w = args.height or 1080, args.width or 1920
            total5 = args.total_frames or 256
            metric = "stabilized + motion-blurred frames/sec (1080p, Flow perspective -> Motion Apply bicubic High; BASELINE configs[2])"
        c5 = run_c5(ctx, torch, dist, device, rank, world, use_dist, total5, h, w, args.steps, args.warmup, chain=args.workload)
        if rank == 0:
            line = {"metric": metric,
                    "value": c5["value"], "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                    "ms_per_step": c5["ms_per_step"], "higher_is_better": True, "vs_baseline": None, "dtype": "f32",
                    "data": "synthetic", "config": c5}
            if world > 1:
                line["scaling"] = "strong"
            print(json.dumps(line), flush=True)
        if use_dist:
            dist.destroy_process_group()
        return 0

    h, w = args.height or 1080, args.width or 1920
    if args.frames is not None:
        total, scaling, label = args.frames * world, "weak", f"{args.frames} frames per GPU"
    elif args.total_frames is not None:
        total, scaling, label = args.total_frames, "strong", f"one {args.total_frames}-frame clip"
    elif world == 1:
        total, scaling, label = 256, "weak", "C2: 256-frame clip"
    else:
        # the metric's own workload on every GPU (BASELINE configs[1]: 256 x 1080p per GPU), as ONE clip of 256 x N frames
        # sharded over the ranks: per-GPU work fixed as N grows.  BASELINE configs[3] (C4: one 1024-frame clip, total work
        # fixed) is timed after it and rides on the same line as the `c4` object.
        total, scaling, label = 256 * world, "weak", f"C2-sized shard per GPU: one {256 * world}-frame clip"

    def time_flow(total_frames: int, steps: int, warmup: int) -> dict:
        """warmup + `steps` timed Flow steps on this rank's shard of one `total_frames`-frame clip, between fences;
        max over ranks.  The shard's frames stay allocated in the result (the one-GPU extras re-use them)."""
        start, end = vd.shard_range(total_frames, world, rank)
        n_local = end - start
        halo = 1 if (rank > 0 and n_local > 0) else 0
        frames = synth_clip(n_local + halo, start - halo, h, w, device)
        torch.cuda.synchronize()
        stats: dict = {}

        plan_log = {"steps": 0, "used": 0, "mismatched_frames": 0}   # what the speculative device plan did, step by step

        def note_plan(verdict):
            if verdict is not None:
                plan_log["steps"] += 1
                plan_log["used"] += int(bool(verdict.get("used")))
                plan_log["mismatched_frames"] = max(plan_log["mismatched_frames"], int(verdict.get("mismatched_frames", 0)))

        def step():
            if not use_dist:
                context = hm._normalize_video_input(frames)   # F0 at the node boundary (the range sniff rides on the gray pass)
                res = fp._stabilize_frames(context, *FLOW_ARGS, ctx=ctx, keep_on_device=True)
                note_plan(res.device_plan)
                return res.frames, res.masks, res.meta
            out = vd.stabilize_sharded(ctx, frames, total_frames, *FLOW_ARGS, stats=stats, want_meta=(rank == 0))
            note_plan(stats.pop("device_plan", None))
            return out

        def fence():
            if use_dist:
                dist.barrier()
            torch.cuda.synchronize()

        for _ in range(warmup):
            out = step()
            del out
        fence()
        # clears the per-kind totals: only the timed steps below are counted.  Inside the timed steps the library records HIP
        # events around the WARP launches only (the roofline kernel's per-launch time has to come from the timed region);
        # the other stages' times come from the extra pass below: an event pair costs the stream ~10 us, 0.05-0.1 ms per
        # step over all stages (tools/timing_cost.py)
        ctx.set_timing(True, warp_only=True)
        stats.clear()
        plan_log.update(steps=0, used=0, mismatched_frames=0)
        meta = None
        t0 = time.perf_counter()
        for _ in range(steps):
            out = step()
            meta = out[2]
            del out
        fence()
        elapsed = time.perf_counter() - t0
        if use_dist:
            t = torch.tensor([elapsed], device=device if dist.get_backend() == "nccl" else "cpu", dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        # HIP events recorded by the library on the launch stream around every call; summed without host
        # synchronisation inside the timed loop, read here after the closing fence
        warp_total_ms, warp_launches = ctx.kernel_ms_stats("warp")
        stage_ms = {"warp": warp_total_ms / max(warp_launches, 1)}
        device_plan = {"used": plan_log["used"] == plan_log["steps"] and plan_log["steps"] > 0,
                       "mismatched_frames_max_per_step": plan_log["mismatched_frames"]}
        if use_dist:   # the worst rank's count (a re-warped frame is an extra launch inside that rank's step)
            t = torch.tensor([device_plan["mismatched_frames_max_per_step"]], device=device if dist.get_backend() == "nccl" else "cpu",
                             dtype=torch.int64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            device_plan["mismatched_frames_max_per_step"] = int(t.item())
        # where DIS goes (HIP events around its stages; collected in ONE extra pass outside the timed region: the events
        # sit between dependent kernels of the coarse-to-fine chain, where they would lengthen the timed steps)
        ctx.set_timing(True)
        for _ in range(3):   # gray / DIS / fit: HIP events around every stage, three passes outside the timed region
            out = step()
            del out
        torch.cuda.synchronize()
        for kind in ("gray", "dis", "fit"):
            total_ms, launches = ctx.kernel_ms_stats(kind)
            stage_ms[kind] = total_ms / max(launches, 1)
        ctx.set_timing(True, detail=True)
        out = step()
        del out
        torch.cuda.synchronize()
        dis_ms = ctx.dis_stage_ms()
        ctx.set_timing(True)
        return {"elapsed": elapsed, "n_local": n_local, "frames": frames, "meta": meta, "step": step, "stage_ms": stage_ms,
                "host_ms": {k: round(v / steps, 3) for k, v in stats.items() if isinstance(v, (int, float))},
                "device_plan": device_plan, "dis_ms": dis_ms}

    def same_clip_on_one_gpu(total_frames: int):
        """the denominator of a strong-scaling ratio is the SAME clip on one GPU, not the N=1 default (C2, 256 frames)"""
        refs = sorted((ROOT / "profiles").glob("r*_c4_single_gpu.json"))
        if not refs:
            return None
        rj = json.loads(refs[-1].read_text())
        if rj.get("total_frames") != total_frames or rj.get("size") != [w, h]:
            return None
        return {"frames_per_s": rj["frames_per_s"], "ms_per_step": rj["ms_per_step"],
                "source": f"profiles/{refs[-1].name} (static: `bench.py --gpus 1 --total-frames {total_frames}` on one MI355X, "
                          "not measured in this run)"}

    run = time_flow(total, args.steps, args.warmup)
    elapsed, n_local, frames, meta, step, stage_ms = (run[k] for k in ("elapsed", "n_local", "frames", "meta", "step", "stage_ms"))
    del run["frames"], run["step"]   # `frames` and `step` (whose closure holds the clip) are the only references left

    rc = 0
    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = total * args.steps / elapsed
        out_w, out_h = meta["stabilization_warp"]["output_size"]
        warp_avg_ms = stage_ms["warp"]
        launch_bytes = WARP_BYTES_PER_PIXEL * out_w * out_h * n_local
        achieved = launch_bytes / (warp_avg_ms * 1e-3) / 1e9
        traffic, traffic_source = warp_traffic_record(n_local, w, h)
        config = {
            "workload": f"{label}, {w}x{h}, Video Stabilizer Flow (DIS) similarity + crop_and_pad, defaults (strength 0.7, "
                        "smooth 0.5, 16 fps), device-resident in/out, entered at the node's input adaptation",
            "frames_per_gpu": n_local,
            "total_frames": total,
            "sharding": "single GPU" if world == 1 else f"contiguous frame shards x{world}, 1-frame halo, RCCL all-gather of fit records",
            "stage_ms": {k: round(float(v), 3) for k, v in stage_ms.items()},
            "stage_ms_note": "warp: HIP events inside the timed steps; gray / dis / fit: three extra passes after them (events cost stream time)",
            "dis_ms": run["dis_ms"],
            "device_plan": run["device_plan"],
        }
        if world > 1 and scaling == "strong" and same_clip_on_one_gpu(total) is not None:
            config["same_clip_on_one_gpu"] = same_clip_on_one_gpu(total)
        if args.rehearse_on_one_gpu:
            config["rehearsal"] = "all ranks share cuda:0, gloo control plane: a code-path rehearsal, NOT a multi-GPU measurement"
        if use_dist:
            config["rank0_host_ms"] = run["host_ms"]
            config["rank0_host_ms_note"] = ("host wall-clock per phase of rank 0's step; gather_fits + gather_counts = the two "
                                            "collectives, plan + meta = replicated host work (meta on rank 0 only)")
        line = {
            "metric": "stabilized frames/sec (1080p, similarity mode)",
            "value": round(value, 2),
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": config,
            "roofline": {
                "bound": "hbm",
                "kernel": "warp_kernel<bilinear,q5,mask>",
                "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": traffic,
                "traffic_source": traffic_source,
                "launch_ms": round(warp_avg_ms, 4),
                "algorithmic_bytes_per_launch": launch_bytes,
            },
        }
        if run["device_plan"]["mismatched_frames_max_per_step"] > 0:
            # frames whose device-plan matrix differed from the host's were warped again: the warp time above is an average
            # over launches of different sizes, which is not the dominant kernel's launch time
            line["roofline"].update(achieved=None, frac=None, launch_ms=None,
                                    note="frames were re-warped inside the timed steps (config.device_plan): the per-launch warp "
                                         "time is a mixed average, no roofline figure is derived from it")
        if world > 1:
            line["scaling"] = scaling      # means nothing on a one-GPU line
        if world == 1 and not use_dist:
            cam = camera_matrices(n_local, 0, w, h)
            work = hm._working_estimation_size(w, h)
            checks = not args.no_checks
            if checks:
                chk_frames, chk_masks, chk_meta = step()   # one more pass, outside the timed region, kept for the checks
                hip_t = [t["matrix"] for t in chk_meta["estimated_motion"]["per_transition"]]
                line["accuracy"] = {"note": "reported transitions vs the clip's analytic M_{i+1} M_i^-1, errors in px at working "
                                            "resolution (centre / worst corner displacement) and max |delta| of the 2x2 part",
                                    "hip": transition_accuracy(hip_t, cam, (w, h), work)}
                try:
                    line["batch_invariance"] = check_batch_invariance(fp, hm, ctx, frames, chk_meta, chk_frames)
                except Exception as exc:
                    line["batch_invariance"] = {"error": f"{type(exc).__name__}: {exc}"}
            if args.cpu_frames >= 2:
                threads = min(16, len(os.sched_getaffinity(0)))
                n_cpu = min(args.cpu_frames, n_local)
                sample = frames[:n_cpu].cpu().numpy()
                port_line, port = cpu_baseline(sample, threads, keep_outputs=(checks and n_cpu == n_local))
                all_cores = len(os.sched_getaffinity(0))
                if all_cores > threads:   # the same sample on every core this process may use (16 is a convention, not a limit)
                    try:
                        wide, _ = cpu_baseline(sample, all_cores)
                        port_line["all_cores"] = {k: wide[k] for k in ("value", "unit", "cores", "stage_s") if k in wide}
                    except Exception as exc:
                        port_line["all_cores"] = {"error": f"{type(exc).__name__}: {exc}"}
                # the real-OpenCV tier, decided at run time (no cv2 in this image or on the GPU box today: "absent")
                try:
                    leg = cv2_leg(ctx, torch, frames, threads, args.cv2_frames)
                except Exception as exc:
                    leg = {"probe": {"cv2": "error", "error": f"{type(exc).__name__}: {exc}"}}
                if "baseline" in leg:
                    line["cpu_baseline"] = leg["baseline"]
                    line["cpu_baseline_port"] = port_line
                    line["cv2_parity"] = leg["parity"]
                else:
                    line["cpu_baseline"] = port_line
                line["cpu_baseline"]["cv2"] = leg["probe"]["cv2"]
                if leg["probe"].get("error") or leg["probe"].get("import_error"):
                    line["cpu_baseline"]["cv2_error"] = leg["probe"].get("error") or leg["probe"].get("import_error")
                if port is not None:
                    line["accuracy"]["cpu_port"] = transition_accuracy(port["transitions"], cam, (w, h), work)
                    port["source"] = sample
                    try:
                        line["parity_at_size"] = check_against_oracle(chk_meta, chk_frames, chk_masks, port)
                    except Exception as exc:
                        line["parity_at_size"] = {"error": f"{type(exc).__name__}: {exc}"}
                del sample, port
            if checks:
                del chk_frames, chk_masks
            if not args.no_extras:
                try:
                    host = frames.cpu()
                    line["host_roundtrip"] = measure_host_roundtrip(nodes, host)
                    # the same clip as a ComfyUI IMAGE decoded from 8-bit video holds it: float32(k) / 255 (the clip crosses PCIe
                    # as bytes: vstab_upload_f32_coded)
                    host.mul_(255.0).round_().clamp_(0.0, 255.0).div_(255.0)
                    line["host_roundtrip_8bit_source"] = measure_host_roundtrip(nodes, host)
                    del host
                    del frames, step
                    torch.cuda.empty_cache()
                    line["motion_apply"] = measure_motion_apply(ctx, torch, device, check=checks and args.cpu_frames >= 2)
                except Exception as exc:  # the headline line must survive a failure of the extras
                    line["extras_error"] = f"{type(exc).__name__}: {exc}"
    c4 = c5 = None
    if use_dist and not args.no_extras:
        # BASELINE configs[3] and configs[4] on the same ranks, outside the timed loop above (every rank takes part: both
        # have collectives).  `--gpus 1 --force-dist` rehearses exactly this path on one GPU's share of the clips.
        del frames, step
        torch.cuda.empty_cache()
        if world > 1 and args.frames is None and args.total_frames is None:
            try:
                c4_run = time_flow(1024, args.steps, max(1, args.warmup))
                c4 = {"workload": f"C4: one 1024-frame clip, {w}x{h}, Flow similarity + crop_and_pad, sharded over {world} GPUs "
                                  "(BASELINE configs[3]; total work fixed)",
                      "scaling": "strong", "value": round(1024 * args.steps / c4_run["elapsed"], 2), "unit": "frames/s",
                      "ms_per_step": round(c4_run["elapsed"] / args.steps * 1e3, 3), "steps": args.steps,
                      "frames_per_gpu": c4_run["n_local"], "stage_ms": {k: round(float(v), 3) for k, v in c4_run["stage_ms"].items()},
                      "rank0_host_ms": c4_run["host_ms"], "device_plan": c4_run["device_plan"], "dis_ms": c4_run["dis_ms"],
                      "same_clip_on_one_gpu": same_clip_on_one_gpu(1024)}
                del c4_run
            except Exception as exc:   # every rank reaches the same collectives or none: a failure here is the same on all ranks
                c4 = {"error": f"{type(exc).__name__}: {exc}"}
            torch.cuda.empty_cache()
        try:
            c5 = run_c5(ctx, torch, dist, device, rank, world, True, args.c5_frames or (512 if world > 1 else 64), 2160, 3840,
                        max(2, args.steps // 2), 1)
        except Exception as exc:   # run_c5 agrees on allocation and on its first pass across the ranks before it times anything
            c5 = {"error": f"{type(exc).__name__}: {exc}"}
    if rank == 0:
        if c4 is not None:
            line["c4"] = c4
            one = c4.get("same_clip_on_one_gpu") if "error" not in c4 else None
            # ">= 6x at 8 GPUs" is a C4 figure (one 1024-frame clip, total work fixed) against the SAME clip on one
            # GPU -- not the weak-scaling `value` of this line
            line["target_check"] = {"config": "C4 (BASELINE configs[3]), strong scaling", "target_speedup_at_8_gpus": 6.0, "n_gpus": world,
                                    "speedup_vs_same_clip_one_gpu": round(one["ms_per_step"] / c4["ms_per_step"], 3) if one else None,
                                    "met": (bool(one["ms_per_step"] / c4["ms_per_step"] >= 6.0) if world == 8 else None) if one else None,
                                    "note": "the >= 6x criterion is evaluated on C4 at 8 GPUs; `value` above is weak scaling (256 frames per GPU)"}
        if c5 is not None:
            line["c5"] = c5
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.destroy_process_group()
    return rc


if __name__ == "__main__":
    sys.exit(main())
