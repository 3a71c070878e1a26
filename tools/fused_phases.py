"""Per-phase time of one workgroup of level_kernel<LEVEL_FUSED> at every pyramid level.

Needs a developer build of libvstab (csrc: `rm build/vstab_dis.o; make EXTRA=-DVSTAB_FUSED_TRACE`), which accumulates
wall_clock64 (100 MHz) differences between the phase barriers of workgroup 7.
"""
import ctypes as C, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
import __graft_entry__ as graft
graft.load_package()
import bench
from vstab_amd import native

ctx = native.Context(0); ctx.set_timing(True)
frames = bench.synth_clip(256, 0, 1080, 1920, torch.device("cuda", 0))
gray = ctx.gray_downscale(frames, (960, 540))
dbg = torch.zeros(16 * 8, dtype=torch.int64, device="cuda")
ctx.lib.vstab_dis_dbg.argtypes = [C.c_void_p]
names = ["densify", "warp", "deriv1+2", "s1 weights", "s2 system", "s3 sor", "s4 write", "merge", "upsample"]
for rep in range(2):
    dbg.zero_()
    ctx.lib.vstab_dis_dbg(C.c_void_p(dbg.data_ptr()))
    ctx.dis_flow_batch(gray, sample_step=8, want_full=False, want_grid=True)
    torch.cuda.synchronize()
    d = dbg.cpu().numpy().reshape(8, 16) / 100.0
    print("dis stage ms %.2f" % ctx.last_kernel_ms("dis"))
    for lvl in range(8):
        if d[lvl].sum() > 0:
            print(" level", lvl, "total us %.0f:" % d[lvl, :9].sum(), ", ".join(f"{n} {v:.0f}" for n, v in zip(names, d[lvl, :9])))
    t = d[7]
    if t.sum() > 0:
        print(" pyramid tail us: load %.0f, levels %s, coarsest prep %.0f" % (t[0], [round(float(v)) for v in t[1:6] if v > 0], t[6]))
