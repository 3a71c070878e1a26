"""Are the 2 x P workgroups of the finest-level patch-inverse-search kernel co-resident?

Needs a developer build of libvstab (csrc: `make clean && make EXTRA=-DVSTAB_PIS_TRACE`), which stamps the
start and end of every wavefront with wall_clock64 (100 MHz).  Prints when workgroups start relative to the first one.
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
dbg = torch.zeros(510 * 16 * 2, dtype=torch.int64, device="cuda")
ctx.lib.vstab_pis_dbg.argtypes = [C.c_void_p]
for rep in range(2):
    dbg.zero_()
    ctx.lib.vstab_pis_dbg(C.c_void_p(dbg.data_ptr()))
    ctx.dis_flow_batch(gray, sample_step=8, want_full=False, want_grid=True)
    torch.cuda.synchronize()
    d = dbg.cpu().numpy().reshape(510, 16, 2)
    t0, t1 = d[..., 0], d[..., 1]
    start = t0.min()
    dur = (t1 - t0) / 100.0
    late = (t0.min(1) - start) / 100.0
    print(f"wave duration us: mean {dur.mean():.0f} p99 {np.percentile(dur, 99):.0f} max {dur.max():.0f}; "
          f"workgroups starting > 100 us after the first: {int((late > 100).sum())} of 510; last end {((t1.max() - start) / 100):.0f} us; "
          f"dis stage {ctx.last_kernel_ms('dis'):.2f} ms")
