"""Rates of the node boundary's transfers: t.to(device) / t.cpu() (runtime staging, one thread) against
vstab_upload / vstab_download (pinned ring + host thread team) for a 256 x 1080p clip; VSTAB_XFER_THREADS sweep."""
import os, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import __graft_entry__ as graft
graft.load_package()
from vstab_amd import native

ctx = native.Context(0)
frames = torch.rand((256, 1080, 1920, 3), dtype=torch.float32)
gb = frames.numel() * 4 / 1e9


def timed(fn, reps=3):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); out = fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0); del out
    return best


d = frames.cuda()
print(f"clip {gb:.2f} GB")
t = timed(lambda: frames.to("cuda")); print(f"torch .to(cuda)     {t*1e3:7.1f} ms  {gb/t:5.1f} GB/s")
t = timed(lambda: d.cpu()); print(f"torch .cpu()        {t*1e3:7.1f} ms  {gb/t:5.1f} GB/s")
for threads in (1, 2, 4, 8, 12, 16):
    os.environ["VSTAB_XFER_THREADS"] = str(threads)
    tu = timed(lambda: ctx.upload(frames)); td = timed(lambda: ctx.download(d))
    print(f"vstab threads={threads:2d}  upload {tu*1e3:7.1f} ms {gb/tu:5.1f} GB/s   download {td*1e3:7.1f} ms {gb/td:5.1f} GB/s")
