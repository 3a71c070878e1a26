"""Kernel time of the two Motion Apply blur configurations (C3 kind: 1080p bicubic S=17; C5 kind: 4K bilinear expand S=33),
device-resident, HIP-event time of the blur launch: python tools/blur_timing.py [frames_1080p frames_4k]"""
import json, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import torch
import __graft_entry__ as graft
graft.load_package()
import bench
from vstab_amd import apply_pipeline as ap, host_math as hm, native

n1 = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n4 = int(sys.argv[2]) if len(sys.argv) > 2 else 16
ctx = native.Context(0)
ctx.set_timing(True)
for fixture, n, h, w, framing, interp, samples in (("shake_c3_256x1080p.json", n1, 1080, 1920, "crop_and_pad", "bicubic", 17),
                                                   ("shake_c5_64x4k.json", n4, 2160, 3840, "expand", "bilinear", 33)):
    meta = {"motion_meta": json.loads((ROOT / "tests" / "golden" / fixture).read_text())}
    blk = meta["motion_meta"]
    blk["per_frame"] = blk["per_frame"][:n]
    blk["frame_count"] = n
    frames = bench.synth_clip(n, 0, h, w, torch.device("cuda", 0))
    for rep in range(4):
        if rep == 1:
            ctx.set_timing(True)
        r = ap.apply_motion(hm._normalize_video_input(frames), meta, (127, 127, 127), framing_mode=framing, interpolation=interp,
                            motion_blur=0.5, motion_blur_samples=samples, ctx=ctx, keep_on_device=True)
        shape = tuple(r.frames.shape)
        del r
    torch.cuda.synchronize()
    ms, launches = ctx.kernel_ms_stats("warp_blur")
    ms /= max(launches, 1)
    print(f"{interp} S={samples} {n}x{w}x{h}: {ms:.3f} ms per launch, {shape[0] * shape[1] * shape[2] * samples / ms / 1e6:.1f} G pixel-samples/s", flush=True)
    del frames
