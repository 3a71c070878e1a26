"""Device time of plan_kernel alone (the replicated plan of a sharded clip) as the clip grows: the fit table of one 256-frame clip
tiled to `total - 1` pairs on the device, `vstab_flow_plan_device` timed with events.   python tools/plan_kernel_timing.py [totals...]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
import __graft_entry__ as graft
graft.load_package()
import bench
from vstab_amd import flow_pipeline as fp, host_math as hm, native
ctx = native.Context(0)
n, h, w = 256, 1080, 1920
dev = torch.device("cuda", 0)
frames = bench.synth_clip(n, 0, h, w, dev)
work = hm._working_estimation_size(w, h)
table = fp.estimate_transitions(ctx, frames, work, "similarity")
for total in [int(a) for a in sys.argv[1:]] or [256, 512, 1024, 2048, 4096]:
    big = np.ascontiguousarray(np.concatenate([table] * (total // 255 + 1))[: total - 1])
    d = torch.from_numpy(big.view(np.uint8).reshape(-1)).to(dev)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    for rep in range(3):
        torch.cuda.synchronize()
        ev[0].record()
        for _ in range(20):
            ctx.flow_plan_device(d.data_ptr(), total - 1, "similarity", (w, h), work, 0.5, 16.0, 0.7, False)
        ev[1].record()
        torch.cuda.synchronize()
    print(f"{total} frames: plan_kernel {ev[0].elapsed_time(ev[1]) / 20 * 1e3:.1f} us per launch (back to back, incl. launch gaps)", flush=True)
