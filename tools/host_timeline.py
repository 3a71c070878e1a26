"""Host timeline of one C2 bench step: every native.Context call with its enter / exit time relative to the step's start, the
step's end, and the time to the next step's first launch -- where the GPU idles between steps while Python works.
    python tools/host_timeline.py"""
import sys, time, types
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
import __graft_entry__ as graft
graft.load_package()
import bench
from vstab_amd import flow_pipeline as fp, host_math as hm, native
ctx = native.Context(0)
n, h, w = 256, 1080, 1920
frames = bench.synth_clip(n, 0, h, w, torch.device("cuda", 0))
log = []
for name in dir(native.Context):
    fn = getattr(native.Context, name)
    if name.startswith("_") or not callable(fn) or name in ("use_torch_stream",):
        continue
    def wrap(fn=fn, name=name):
        def inner(self, *a, **k):
            t0 = time.perf_counter()
            try:
                return fn(self, *a, **k)
            finally:
                log.append((name, t0, time.perf_counter()))
        return inner
    setattr(native.Context, name, wrap())

def step():
    context = hm._normalize_video_input(frames)
    res = fp._stabilize_frames(context, *bench.FLOW_ARGS, ctx=ctx, keep_on_device=True)
    return res.frames, res.masks, res.meta

for _ in range(5):
    out = step(); del out
torch.cuda.synchronize()
rows = []
for rep in range(8):
    log.clear()
    t0 = time.perf_counter()
    out = step()
    t1 = time.perf_counter()
    del out
    t2 = time.perf_counter()
    rows.append((t0, t1, t2, list(log)))
t0, t1, t2, lg = rows[-1]
print(f"step {(t1 - t0) * 1e3:.3f} ms, del out {(t2 - t1) * 1e6:.0f} us")
prev = t0
for name, a, b in lg:
    print(f"  +{(a - t0) * 1e6:8.0f} us  {name:28s} {(b - a) * 1e6:8.0f} us   (python before it {(a - prev) * 1e6:6.0f} us)")
    prev = b
print(f"  +{(t1 - t0) * 1e6:8.0f} us  step returns              (python after the last call {(t1 - prev) * 1e6:6.0f} us)")
