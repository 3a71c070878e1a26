"""Workload for the PMC passes of tools/pmc_dis.sh: two Flow passes (C2, 256 x 1080p) and one Motion Apply pass
(C3 kind: bicubic, blur 0.5, 17 samples, 64 x 1080p) so that level_kernel, pis4_kernel and the blur warp kernel
each appear a few times.  Run directly under rocprofv3 (`-- python3 tools/pmc_target.py`)."""
import json, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import torch
import __graft_entry__ as graft
graft.load_package()
import bench
from vstab_amd import apply_pipeline as ap, flow_pipeline as fp, host_math as hm, native

def say(msg):
    print(f"[pmc_target] {msg}", flush=True)


ctx = native.Context(0)


def _settled(t):
    """Input context with the value-range sniff settled up front: Motion Apply then stays on ONE stream (its optimistic sniff
    would run on a side stream; under counter collection cross-stream waits can deadlock, profiles/r03_pmc_stuck_pass.md)."""
    c = hm._normalize_video_input(t)
    hm.resolve_value_range(c)
    return c

frames = bench.synth_clip(256, 0, 1080, 1920, torch.device("cuda", 0))
say("clip ready")
for k in range(2):
    r = fp._stabilize_frames(hm._normalize_video_input(frames), *bench.FLOW_ARGS, ctx=ctx, keep_on_device=True)
    del r
    ctx.synchronize()   # raises VstabError if a kernel left a bit in the device status word (an expired DIS wait)
    say(f"flow pass {k} done, device status clean")
meta = {"motion_meta": json.loads((ROOT / "tests" / "golden" / "shake_c3_256x1080p.json").read_text())}
blk = meta["motion_meta"]
blk["per_frame"] = blk["per_frame"][:64]
blk["frame_count"] = 64
r = ap.apply_motion(_settled(frames[:64]), meta, (127, 127, 127), framing_mode="crop_and_pad", interpolation="bicubic",
                    motion_blur=0.5, motion_blur_samples=17, ctx=ctx, keep_on_device=True)
ctx.synchronize()
say("motion apply pass done, device status clean")
print("done")
