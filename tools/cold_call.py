"""The FIRST call of the Flow node in a fresh process against the calls after it (CPU tensor in -> CPU tensors out, 256 x 1080p, a clip as
decoded from 8-bit video), with every native.Context call of the first one: what a one-clip ComfyUI run pays that a timed loop never sees.
    python tools/cold_call.py"""
import sys, time
t_start = time.perf_counter()
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
import __graft_entry__ as graft
graft.load_package()
from vstab_amd import native, nodes
print(f"imports {time.perf_counter() - t_start:.2f} s")
n, h, w = 256, 1080, 1920
rng = np.random.default_rng(0)
base = rng.integers(0, 256, (h + 64, w + 64, 3), dtype=np.uint8)
base = (base.astype(np.float32) + np.roll(base, 1, 0) + np.roll(base, 1, 1) + np.roll(base, (1, 1), (0, 1))) / 4.0   # some texture
clip = np.empty((n, h, w, 3), np.float32)
for i in range(n):
    dy, dx = int(16 + 12 * np.sin(i / 9.0)), int(16 + 12 * np.cos(i / 7.0))
    clip[i] = np.round(base[dy:dy + h, dx:dx + w]) / np.float32(255.0)
frames = torch.from_numpy(clip)
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
args = (16.0, "crop_and_pad", "similarity", False, 0.7, 0.5, 0.6, "#7F7F7F")
for call in range(4):
    log.clear()
    t0 = time.perf_counter()
    out = nodes.VideoStabilizerFlow.execute(frames, *args)
    dt = time.perf_counter() - t0
    print(f"call {call}: {dt * 1e3:8.1f} ms")
    if call in (0, 3):
        agg = {}
        for name, a, b in log:
            agg[name] = agg.get(name, 0.0) + (b - a)
        inside = sum(b - a for _, a, b in log)
        print("   " + ", ".join(f"{k} {v * 1e3:.1f}" for k, v in sorted(agg.items(), key=lambda kv: -kv[1])[:10]) + f"; outside the library {(dt - inside) * 1e3:.1f} ms")
    del out
