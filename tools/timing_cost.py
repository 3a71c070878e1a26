"""What the library's HIP-event timers cost inside a C2 step: the same 256 x 1080p Flow step timed on the host clock with the
per-stage events on (as bench.py runs it) and off.  python tools/timing_cost.py"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import __graft_entry__ as graft
graft.load_package()
import bench
from vstab_amd import flow_pipeline as fp, host_math as hm, native
ctx = native.Context(0)
frames = bench.synth_clip(256, 0, 1080, 1920, torch.device("cuda", 0))
def loop(n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        r = fp._stabilize_frames(hm._normalize_video_input(frames), *bench.FLOW_ARGS, ctx=ctx, keep_on_device=True); del r
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for rep in range(3):
    for on in (True, False):
        ctx.set_timing(on); loop(3)
        print(f"timing {'on ' if on else 'off'}: {loop(20):.3f} ms per step", flush=True)
