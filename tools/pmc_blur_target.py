"""Workload for tools/pmc_blur_hist.sh: Motion Apply blur passes only -- C3 kind (1080p, bicubic, blur 0.5, 17 samples,
32 frames) and C5 kind (4K, bilinear, expand, 33 samples, 8 frames), two passes each, device-resident."""
import json, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import torch
import __graft_entry__ as graft
graft.load_package()
import bench
from vstab_amd import apply_pipeline as ap, host_math as hm, native

ctx = native.Context(0)


def _settled(t):
    """Input context with the value-range sniff settled up front: Motion Apply then stays on ONE stream (its optimistic sniff
    would run on a side stream; under counter collection cross-stream waits can deadlock, profiles/r03_pmc_stuck_pass.md)."""
    c = hm._normalize_video_input(t)
    hm.resolve_value_range(c)
    return c

for fixture, n, h, w, framing, interp, samples in (("shake_c3_256x1080p.json", 32, 1080, 1920, "crop_and_pad", "bicubic", 17),
                                                   ("shake_c5_64x4k.json", 8, 2160, 3840, "expand", "bilinear", 33)):
    meta = {"motion_meta": json.loads((ROOT / "tests" / "golden" / fixture).read_text())}
    blk = meta["motion_meta"]
    blk["per_frame"] = blk["per_frame"][:n]
    blk["frame_count"] = n
    frames = bench.synth_clip(n, 0, h, w, torch.device("cuda", 0))
    for _ in range(2):
        r = ap.apply_motion(_settled(frames), meta, (127, 127, 127), framing_mode=framing, interpolation=interp,
                            motion_blur=0.5, motion_blur_samples=samples, ctx=ctx, keep_on_device=True)
        shape = tuple(r.frames.shape)
        del r
    ctx.synchronize()
    print(f"[pmc_blur_target] {interp} S={samples} {n}x{w}x{h} -> {shape}: pixel-samples per pass {shape[0] * shape[1] * shape[2] * samples}", flush=True)
    del frames
print("done")
