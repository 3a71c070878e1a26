"""Per-phase time of one workgroup of homography_kernel (perspective fit of the C3 clip's 255 pairs).
Needs a developer build (`tools/build_variant.sh htrace - -DVSTAB_HOMOGRAPHY_TRACE`, loaded through VSTAB_LIB)."""
import ctypes as C, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
import __graft_entry__ as graft
graft.load_package()
import bench
from tests.util import shake_path
from vstab_amd import native, flow_pipeline as fp

ctx = native.Context(0); ctx.set_timing(True)
cam = shake_path(256, 1920, 1080, "perspective", seed=3, amp=1.0)
frames = bench.synth_clip(256, 0, 1080, 1920, torch.device("cuda", 0), mats=cam)
gray = ctx.gray_downscale(frames, (960, 540))
_, grid = ctx.dis_flow_batch(gray, sample_step=8, want_full=False, want_grid=True)
dbg = torch.zeros(16, dtype=torch.int64, device="cuda")
ctx.lib.vstab_homography_dbg.argtypes = [C.c_void_p]
names = ["rng+subset (1 lane)", "16 x 4-pt DLT/Jacobi", "score", "bookkeeping", "refit sums", "refit LtL reduce", "refit Jacobi (1 lane)",
         "LM first accumulate", "LM solve (1 lane)", "LM accumulate (no J)", "LM rho/lambda (1 lane)", "LM accept + accumulate (J)", "residual"]
for rep in range(2):
    dbg.zero_()
    ctx.lib.vstab_homography_dbg(C.c_void_p(dbg.data_ptr()))
    ctx.sample_fit_batch(grid, 8, "perspective")
    torch.cuda.synchronize()
    d = dbg.cpu().numpy()
    print("fit stage ms %.3f   LM iterations of the traced pair: %d" % (ctx.last_kernel_ms("fit"), d[14]))
    print("  " + ", ".join(f"{n} {v / 100.0:.0f}" for n, v in zip(names, d[:13])), " | total us %.0f" % (d[:13].sum() / 100.0))
