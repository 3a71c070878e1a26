"""Motion Apply as ComfyUI calls it (CPU tensor + meta in -> CPU tensors out), C3's settings (bicubic, motion blur 0.5, High = 17 samples)
on a 256 x 1080p clip decoded from 8-bit video: per-call time with every native.Context call of the last one.
    python tools/apply_roundtrip.py"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
import __graft_entry__ as graft
graft.load_package()
import bench
from vstab_amd import native, nodes
n, h, w = 256, 1080, 1920
frames = bench.synth_clip(n, 0, h, w, torch.device("cuda", 0)).cpu()
frames.mul_(255.0).round_().clamp_(0.0, 255.0).div_(255.0)
meta = nodes.VideoStabilizerFlow.execute(frames, 16.0, "crop_and_pad", "perspective", False, 0.7, 0.5, 0.6, "#7F7F7F")[2]
log = []
for name in dir(native.Context):
    fn = getattr(native.Context, name)
    if name.startswith("_") or not callable(fn):
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
for blur, quality in ((0.5, "High"), (0.0, "Standard")):
    for call in range(3):
        log.clear()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = nodes.VideoStabilizerMotionApply.execute(frames, meta, "crop_and_pad", "bicubic", "#7F7F7F", blur, quality)
        dt = time.perf_counter() - t0
        agg = {}
        for name, a, b in log:
            agg[name] = agg.get(name, 0.0) + (b - a)
        print(f"blur {blur} {quality}: {dt * 1e3:7.1f} ms  " + ", ".join(f"{k} {v * 1e3:.1f}" for k, v in sorted(agg.items(), key=lambda kv: -kv[1])[:6]), flush=True)
        del out
