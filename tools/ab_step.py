"""A/B of builds of libvstab.so inside ONE gpurun call: whole C2 step (host clock, events around the warp only, as bench.py
times it) and the per-stage HIP-event times, each library in its own child process, interleaved twice.
    python tools/ab_step.py libA.so libB.so ..."""
import subprocess, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
CHILD = r'''
import sys, time
sys.path.insert(0, %r)
import numpy as np, torch
import __graft_entry__ as graft
graft.load_package()
from vstab_amd import native
native.LIB_PATH = __import__("pathlib").Path(sys.argv[1]).resolve()
import bench
from vstab_amd import flow_pipeline as fp, host_math as hm
ctx = native.Context(0)
frames = bench.synth_clip(256, 0, 1080, 1920, torch.device("cuda", 0))
def step():
    r = fp._stabilize_frames(hm._normalize_video_input(frames), *bench.FLOW_ARGS, ctx=ctx, keep_on_device=True); del r
def loop(n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): step()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
try:
    ctx.set_timing(True, warp_only=True)
except TypeError:
    ctx.set_timing(False)
loop(5)
ms = [loop(20) for _ in range(3)]
ctx.set_timing(True); loop(3)
st = {k: round(ctx.kernel_ms_stats(k)[0] / max(ctx.kernel_ms_stats(k)[1], 1), 3) for k in ("gray", "dis", "fit", "warp")}
print(sys.argv[1].split("/")[-1], "ms/step", [round(m, 3) for m in ms], st)
'''
for _ in range(2):
    for lib in sys.argv[1:]:
        out = subprocess.run([sys.executable, "-c", CHILD % str(ROOT), str(ROOT / lib)], capture_output=True, text=True)
        print(out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-600:], flush=True)
