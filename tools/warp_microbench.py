"""Micro-benchmark of the warp kernels (device-resident), prints achieved algorithmic GB/s."""
import argparse
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import __graft_entry__ as graft

graft.load_package()
from vstab_amd import native  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=64)
ap.add_argument("--h", type=int, default=1080)
ap.add_argument("--w", type=int, default=1920)
ap.add_argument("--interp", default="bilinear")
ap.add_argument("--blur", type=int, default=0, help="samples (0 = plain warp)")
ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--persp", action="store_true")
ap.add_argument("--no-count", action="store_true", help="no padded-pixel count (Motion Apply's call shape; bicubic then runs the staged kernel)")
args = ap.parse_args()

ctx = native.default_context()
ctx.set_timing(True)
n, h, w = args.n, args.h, args.w
frames = torch.rand((n, h, w, 3), device="cuda", dtype=torch.float32)
rng = np.random.default_rng(0)
mats = np.tile(np.eye(3), (n, 1, 1))
th = rng.uniform(-0.01, 0.01, n)
sc = rng.uniform(0.99, 1.01, n)
mats[:, 0, 0] = sc * np.cos(th); mats[:, 0, 1] = -sc * np.sin(th)
mats[:, 1, 0] = sc * np.sin(th); mats[:, 1, 1] = sc * np.cos(th)
mats[:, 0, 2] = rng.uniform(-20, 20, n); mats[:, 1, 2] = rng.uniform(-12, 12, n)
if args.persp:
    mats[:, 2, 0] = rng.uniform(-1e-5, 1e-5, n); mats[:, 2, 1] = rng.uniform(-1e-5, 1e-5, n)
border = np.array([127, 127, 127], np.float32) / 255
dst = torch.empty((n, h, w, 3), device="cuda")
mask = torch.empty((n, h, w), device="cuda")
times = []
for r in range(args.reps + 2):
    if args.blur:
        ctx.warp_blur_batch(frames, mats, (w, h), 0.5, args.blur, interp=args.interp, border=border, out=dst, out_mask=mask)
        ms = ctx.last_kernel_ms("warp_blur")
    else:
        ctx.warp_batch(frames, mats.astype(np.float32), (w, h), interp=args.interp, border=border, want_count=not args.no_count, out=dst, out_mask=mask)
        ms = ctx.last_kernel_ms("warp")
    if r >= 2:
        times.append(ms)
ms = float(np.median(times))
bytes_alg = n * h * w * 28
print(f"{args.interp} blur={args.blur} persp={args.persp}: {ms:.3f} ms for {n} frames -> {n/ms*1e3:.0f} frames/s, "
      f"{bytes_alg/ms/1e6:.1f} GB/s algorithmic ({bytes_alg/ms/1e6/8000*100:.1f}% of 8 TB/s); min {min(times):.3f} ms")
