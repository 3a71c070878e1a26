#!/usr/bin/env python3
"""Oracle-independent accuracy of the Flow node on analytic clips (bench.synth_clip: frame_i(p) = T(M_i^-1 p)):

  * per-transition error distribution against M_{i+1} M_i^-1 for several sizes / modes / motion magnitudes
  * camera_lock + strength 1 on a translation-only path: the output against the static texture it should show

Prints one JSON object per case; the numbers back the bounds written in tests/test_analytic_gpu.py.

    python tools/analytic_accuracy.py [--frames 64]
"""
import argparse
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

import bench  # noqa: E402
import __graft_entry__ as graft  # noqa: E402


from tests.util import shake_path  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=64)
    args = ap.parse_args()
    import torch

    graft.load_package()
    from vstab_amd import flow_pipeline as fp
    from vstab_amd import host_math as hm
    from vstab_amd import native

    ctx = native.default_context()
    dev = torch.device("cuda", 0)
    n = args.frames
    for (w, h) in ((960, 540), (1920, 1080), (540, 960)):
        for kind in ("translation", "similarity", "perspective"):
            for amp in (0.25, 1.0, 3.0):
                cam = shake_path(n, w, h, kind, amp=amp)
                frames = bench.synth_clip(n, 0, h, w, dev, mats=cam)
                res = fp._stabilize_frames(hm._normalize_video_input(frames), "crop_and_pad", kind, False, 0.7, 0.5, 0.6,
                                           (127, 127, 127), 16.0, ctx=ctx, keep_on_device=True)
                tr = res.meta["estimated_motion"]["per_transition"]
                acc = bench.transition_accuracy([t["matrix"] for t in tr], cam, (w, h), hm._working_estimation_size(w, h))
                modes = sorted({t["mode"] for t in tr})
                print(json.dumps({"case": f"{w}x{h} {kind} amp {amp}", "modes": modes, **acc}), flush=True)
                del frames, res
    # camera lock: translation-only path, strength 1 -> every output frame shows frame 0's view (shifted by the recentring)
    for (w, h) in ((960, 540), (1920, 1080)):
        for mode in ("translation", "similarity"):
            cam = shake_path(n, w, h, "translation", amp=1.0)
            frames = bench.synth_clip(n, 0, h, w, dev, mats=cam)
            res = fp._stabilize_frames(hm._normalize_video_input(frames), "crop_and_pad", mode, True, 1.0, 0.5, 0.6,
                                       (127, 127, 127), 16.0, ctx=ctx, keep_on_device=True)
            print(json.dumps({"case": f"camera_lock {w}x{h} {mode}", **bench.static_texture_error(res, cam, frames, dev)}), flush=True)
            del frames, res


if __name__ == "__main__":
    main()
