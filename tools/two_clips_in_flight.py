"""Throughput with TWO clips in flight on one GPU (two host threads, two library contexts, two HIP streams): does the
HBM-bound half of one clip's step (gray, warp) overlap the VALU-bound half (DIS) of the other's?
     python tools/two_clips_in_flight.py [steps]"""
import sys, threading, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
import __graft_entry__ as graft
graft.load_package()
import bench
from vstab_amd import flow_pipeline as fp, host_math as hm, native

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda", 0)
clips = [bench.synth_clip(256, 0, 1080, 1920, dev, seed=1234 + i) for i in range(2)]


def worker(i, n, out):
    stream = torch.cuda.Stream(device=dev)
    ctx = native.Context(0)
    with torch.cuda.stream(stream):
        for _ in range(3):
            fp._stabilize_frames(hm._normalize_video_input(clips[i]), *bench.FLOW_ARGS, ctx=ctx, keep_on_device=True)
        stream.synchronize()
        out["ready"].wait()
        t0 = time.perf_counter()
        for _ in range(n):
            r = fp._stabilize_frames(hm._normalize_video_input(clips[i]), *bench.FLOW_ARGS, ctx=ctx, keep_on_device=True)
            del r
        stream.synchronize()
        out[i] = time.perf_counter() - t0


for nthreads in (1, 2, 1, 2):
    out = {"ready": threading.Barrier(nthreads)}
    ths = [threading.Thread(target=worker, args=(i, steps, out)) for i in range(nthreads)]
    t0 = time.perf_counter()
    for t in ths: t.start()
    for t in ths: t.join()
    wall = max(out[i] for i in range(nthreads))
    print(f"{nthreads} clip(s) in flight: {nthreads * steps * 256 / wall:9.0f} frames/s  ({wall / steps * 1e3:.2f} ms per step per thread)", flush=True)
