"""Unprofiled helper of the PMC passes: synthesise the C2 clip ONCE, outside the profiler, and park it under /tmp, so that the
profiled process (tools/pmc_target.py) loads it with a handful of host-to-device copies instead of issuing ~10^4 serialised
torch dispatches under counter collection before the kernels of interest (the three passes killed at their limit
all stalled somewhere between the first synthesis kernel and the first library kernel).
    python3 tools/pmc_make_clip.py [frames=256]  ->  /tmp/vstab_pmc_clip_<frames>.npy (float32 [frames,1080,1920,3])"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import numpy as np, torch
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
out = Path(f"/tmp/vstab_pmc_clip_{n}.npy")
if not out.exists():
    frames = bench.synth_clip(n, 0, 1080, 1920, torch.device("cuda", 0))
    torch.cuda.synchronize()
    tmp = out.with_suffix(".tmp.npy")
    np.save(tmp, frames.cpu().numpy())
    tmp.rename(out)
print(out, out.stat().st_size)
