"""Rates of the node boundary's transfers: t.to(device) / t.cpu() (runtime staging, one thread) against
vstab_upload / vstab_download (pinned ring + host thread team) for a 256 x 1080p clip; VSTAB_XFER_THREADS / THP sweep,
then the Flow node end to end (CPU tensor in -> CPU tensors out)."""
import os, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import __graft_entry__ as graft
graft.load_package()
from vstab_amd import native, nodes
import bench

ctx = native.Context(0)
frames = bench.synth_clip(256, 0, 1080, 1920, torch.device("cuda", 0)).cpu()
gb = frames.numel() * 4 / 1e9
print("cores", len(os.sched_getaffinity(0)), "THP", open("/sys/kernel/mm/transparent_hugepage/enabled").read().strip())


def timed(fn, reps=3):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); out = fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0); del out
    return best


d = frames.cuda()
print(f"clip {gb:.2f} GB")
t = timed(lambda: frames.to("cuda")); print(f"torch .to(cuda)     {t*1e3:7.1f} ms  {gb/t:5.1f} GB/s")
t = timed(lambda: d.cpu()); print(f"torch .cpu()        {t*1e3:7.1f} ms  {gb/t:5.1f} GB/s")
for thp in ("0", "1"):
    os.environ["VSTAB_XFER_THP"] = thp
    for threads in (4, 8, 16):
        os.environ["VSTAB_XFER_THREADS"] = str(threads)
        tu = timed(lambda: ctx.upload(frames)); td = timed(lambda: ctx.download(d))
        print(f"vstab thp={thp} threads={threads:2d}  upload {tu*1e3:7.1f} ms {gb/tu:5.1f} GB/s   download {td*1e3:7.1f} ms {gb/td:5.1f} GB/s")
del d
os.environ.pop("VSTAB_XFER_THREADS"); os.environ.pop("VSTAB_XFER_THP")
# coded forms: a clip as an IMAGE decoded from 8-bit video holds it (float32(k) / 255), a 0 / 1 mask
q = (frames * 255.0).round_().clamp_(0.0, 255.0).div_(255.0)
mask = (torch.rand((256, 1080, 1920), device="cuda") < 0.1).float()
mgb = mask.numel() * 4 / 1e9
for coded in ("0", "1"):
    os.environ["VSTAB_XFER_CODED"] = coded
    tu = timed(lambda: ctx.upload(q)); info = ctx.last_upload_coded
    tm = timed(lambda: ctx.download(mask, mask=True))
    print(f"coded={coded}  upload of an 8-bit-sourced clip {tu*1e3:7.1f} ms ({gb/tu:5.1f} GB/s of float32, {info[0]}/{info[1]} chunks as bytes)   "
          f"0/1 mask download {tm*1e3:7.1f} ms ({mgb/tm:5.1f} GB/s of float32, as bytes: {ctx.last_download_coded})")
    tf = timed(lambda: ctx.upload(frames))
    print(f"coded={coded}  upload of the float clip {tf*1e3:7.1f} ms ({gb/tf:5.1f} GB/s; {ctx.last_upload_coded[0]} chunks as bytes)")
os.environ.pop("VSTAB_XFER_CODED")
del mask
args = (16.0, "crop_and_pad", "similarity", False, 0.7, 0.5, 0.6, "#7F7F7F")
for name, clip in (("float clip", frames), ("8-bit-sourced clip", q)):
    for coded in ("0", "1"):
        os.environ["VSTAB_XFER_CODED"] = coded
        for rep in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            out = nodes.VideoStabilizerFlow.execute(clip, *args)
            dt = time.perf_counter() - t0
            print(f"Flow node, CPU in -> CPU out, {name}, coded={coded}: {dt*1e3:.1f} ms = {256/dt:.0f} frames/s")
            del out
