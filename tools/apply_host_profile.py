"""cProfile of the HOST side of Motion Apply's launch (C3's and C5's settings, device-resident frames): what runs between the Flow node's
return and the blur warp's launch, while the GPU idles in a Flow -> Motion Apply chain.   python tools/apply_host_profile.py [c3|c5]"""
import cProfile, pstats, sys, time
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import __graft_entry__ as graft
graft.load_package()
import bench
from vstab_amd import apply_pipeline as ap, flow_pipeline as fp, host_math as hm, native
which = sys.argv[1] if len(sys.argv) > 1 else "c3"
FLOW_ARGS, APPLY, tag, what = bench.CHAINS[which]
n, h, w = (256, 1080, 1920) if which == "c3" else (64, 2160, 3840)
ctx = native.Context(0)
frames = bench.synth_clip(n, 0, h, w, torch.device("cuda", 0))
meta = fp._stabilize_frames(hm._normalize_video_input(frames), *FLOW_ARGS, ctx=ctx, keep_on_device=True).meta
def launch():
    t0 = time.perf_counter()
    r = ap.apply_motion(hm._normalize_video_input(frames), meta, (127, 127, 127), ctx=ctx, keep_on_device=True, **APPLY)
    dt = time.perf_counter() - t0
    torch.cuda.synchronize()
    return dt
for _ in range(3): launch()
print(which, "host ms until the launch returns:", [round(launch() * 1e3, 3) for _ in range(5)])
pr = cProfile.Profile(); pr.enable()
for _ in range(10): launch()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
