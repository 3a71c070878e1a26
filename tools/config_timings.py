"""Device-resident timings of the other BASELINE configs (not bench lines): C3 and the per-GPU share of C5."""
import argparse, sys, time
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import __graft_entry__ as graft
graft.load_package()
import bench
from vstab_amd import apply_pipeline as ap, flow_pipeline as fp, host_math as hm, native
import os
if os.environ.get("VSTAB_LIB"):   # A/B runs of two builds
    native.LIB_PATH = Path(os.environ["VSTAB_LIB"]).resolve()

ap_ = argparse.ArgumentParser(); ap_.add_argument("--config", default="c3"); ap_.add_argument("--reps", type=int, default=3)
args = ap_.parse_args()
ctx = native.Context(0); ctx.set_timing(True)
if args.config == "c1":
    # BASELINE config C1: 64 x 480p, Classic estimator, translation (the reference runs this one on the CPU)
    n, h, w = 64, 480, 854
    frames = bench.synth_clip(n, 0, h, w, torch.device("cuda", 0))
    mk = lambda: hm.VideoContext([None] * n, hm.FrameAdapter(np.dtype(np.float32), False, "0_1", "torch", False), w, h, 3, None, "sequence", {}, batch=frames)
    for r in range(args.reps + 1):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        res = fp._stabilize_frames(mk(), "crop_and_pad", "translation", False, 0.7, 0.5, 0.6, (127, 127, 127), 16.0, ctx=ctx,
                                   keep_on_device=True, estimator="classic")
        torch.cuda.synchronize(); t1 = time.perf_counter()
        if r:
            per = res.meta["estimated_motion"]["per_transition"]
            print(f"c1: classic {1e3*(t1-t0):.2f} ms ({n/(t1-t0):.0f} f/s); gftt {ctx.last_kernel_ms('gftt'):.2f} ms, lk {ctx.last_kernel_ms('lk'):.2f} ms, "
                  f"fit {ctx.last_kernel_ms('fit'):.2f} ms, warp {ctx.last_kernel_ms('warp'):.2f} ms; mean confidence {np.mean([t['confidence'] for t in per]):.3f}")
    sys.exit(0)
if args.config == "c3":
    n, h, w, mode, framing, interp, samples = 256, 1080, 1920, "perspective", "crop_and_pad", "bicubic", 17
else:
    n, h, w, mode, framing, interp, samples = 64, 2160, 3840, "similarity", "expand", "bilinear", 33
frames = bench.synth_clip(n, 0, h, w, torch.device("cuda", 0))
def mkctx():
    return hm.VideoContext([None] * n, hm.FrameAdapter(np.dtype(np.float32), False, "0_1", "torch", False), w, h, 3, None, "sequence", {}, batch=frames)
for r in range(args.reps + 1):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    res = fp._stabilize_frames(mkctx(), framing, mode, False, 0.7, 0.5, 0.6, (127, 127, 127), 16.0, ctx=ctx, keep_on_device=True)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    out = ap.apply_motion(mkctx(), res.meta, (127, 127, 127), framing_mode=framing, interpolation=interp, motion_blur=0.5,
                          motion_blur_samples=samples, ctx=ctx, keep_on_device=True)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    if r:
        print(f"{args.config}: flow {1e3*(t1-t0):.1f} ms ({n/(t1-t0):.0f} f/s), motion-apply {1e3*(t2-t1):.1f} ms ({n/(t2-t1):.0f} f/s), "
              f"blur kernel {ctx.last_kernel_ms('warp_blur'):.1f} ms, out {tuple(out.frames.shape)}, modes {set(t['mode'] for t in res.meta['estimated_motion']['per_transition'])}")
    del out
