"""Workload for the PMC passes of tools/pmc_dis.sh / pmc_traffic.sh: two Flow passes (C2, 256 x 1080p) and one Motion Apply
pass (C3 kind: bicubic, blur 0.5, 17 samples, 64 x 1080p) so that level_kernel, pis4_kernel and the blur warp kernel each
appear a few times.  Run directly under rocprofv3 (`-- python3 tools/pmc_target.py`).

Built so that a pass that stalls says WHERE (three passes were killed at their limit in rounds 2-4; their logs ended at "clip
ready", printed after the clip's kernels were QUEUED, not finished -- so the stall was somewhere between the first synthesis
kernel and the first library kernel, location unknown; profiles/r04_pmc_failed_pass/README.md):
  * the context is created and ONE trivial library kernel is launched and waited for before anything else runs: the
    library's code object is loaded, and its first dispatch is profiled, while nothing else is in flight;
  * the clip comes from /tmp (tools/pmc_make_clip.py, run unprofiled by tools/pmc_lib.sh) through a few large copies; only if
    that file is missing is it synthesised here, with a synchronisation and a line every 32 frames;
  * every stage ends in a host synchronisation and a line;
  * faulthandler dumps every thread's Python stack after 150 s without finishing, i.e. before the pass's 200 s limit."""
import faulthandler, json, sys
from pathlib import Path
faulthandler.dump_traceback_later(150, exit=False)
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import numpy as np
import torch
import __graft_entry__ as graft
graft.load_package()
import bench
from vstab_amd import apply_pipeline as ap, flow_pipeline as fp, host_math as hm, native

def say(msg):
    print(f"[pmc_target] {msg}", flush=True)


ctx = native.Context(0)
tiny = torch.zeros((2, 8, 16, 3), dtype=torch.float32, device="cuda")
torch.cuda.synchronize(); say("device up, torch's first kernels done")
ctx.frame_range(tiny); ctx.synchronize(); say("first library kernel done (code object loaded)")


def _settled(t):
    """Input context with the value-range sniff settled up front: Motion Apply then stays on ONE stream (its optimistic sniff
    would run on a side stream; under counter collection cross-stream waits can deadlock, profiles/r03_pmc_stuck_pass.md)."""
    c = hm._normalize_video_input(t)
    hm.resolve_value_range(c)
    return c

N = 256
parked = Path(f"/tmp/vstab_pmc_clip_{N}.npy")
if parked.exists():
    host = np.load(parked, mmap_mode="r")
    frames = torch.empty((N, 1080, 1920, 3), dtype=torch.float32, device="cuda")
    for a in range(0, N, 32):
        frames[a:a + 32].copy_(torch.from_numpy(np.ascontiguousarray(host[a:a + 32])))
    torch.cuda.synchronize(); say("clip loaded from /tmp (8 copies, no synthesis kernels under the profiler)")
else:
    parts = []
    for a in range(0, N, 32):
        parts.append(bench.synth_clip(32, a, 1080, 1920, torch.device("cuda", 0)))
        torch.cuda.synchronize(); say(f"clip synthesised up to frame {a + 32}")
    frames = torch.cat(parts); del parts
    torch.cuda.synchronize(); say("clip synthesised")
work = hm._working_estimation_size(1920, 1080)
gray = ctx.gray_downscale(frames, work); ctx.synchronize(); say("stage gray done")
_, grid = ctx.dis_flow_batch(gray, sample_step=fp.SAMPLE_STEP, want_full=False, want_grid=True); ctx.synchronize(); say("stage dis done")
table = ctx.sample_fit_batch(grid, fp.SAMPLE_STEP, "similarity"); say("stage fit done")
plan = fp.plan_stabilization(ctx, table, (1920, 1080), 256, "crop_and_pad", "similarity", False, 0.7, 0.5, 0.6, (127, 127, 127), 16.0, 16.0)
d_, m_, c_ = ctx.warp_batch(frames, plan.final_matrices, (1920, 1080), border=hm.border_value((127, 127, 127)), want_mask=True, want_count=True)
ctx.synchronize(); say("stage warp done")
del gray, grid, d_, m_, c_
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
faulthandler.cancel_dump_traceback_later()
print("done")
