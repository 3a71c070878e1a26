"""Device-resident timing of every node / mode combination on one clip: a sweep for performance pathologies."""
import itertools, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
import __graft_entry__ as graft
graft.load_package()
import bench
from vstab_amd import apply_pipeline as ap, flow_pipeline as fp, host_math as hm, native

ctx = native.Context(0)
n, h, w = 64, 1080, 1920
frames = bench.synth_clip(n, 0, h, w, torch.device("cuda", 0))
mk = lambda: hm.VideoContext([None] * n, hm.FrameAdapter(np.dtype(np.float32), False, "0_1", "torch", False), w, h, 3, None, "sequence", {}, batch=frames)

def timed(fn, reps=2):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps): out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps * 1e3, out

metas = {}
for est, framing, mode in itertools.product(("flow", "classic"), ("crop_and_pad", "expand", "crop"), ("translation", "similarity", "perspective")):
    ms, res = timed(lambda: fp._stabilize_frames(mk(), framing, mode, False, 0.7, 0.5, 0.6, (127, 127, 127), 16.0, ctx=ctx, keep_on_device=True, estimator=est))
    metas[(est, framing, mode)] = res.meta
    print(f"{est:8s} {framing:13s} {mode:12s} {ms:8.2f} ms   applied={res.meta['transform_mode_applied']}")
    del res
meta = metas[("flow", "crop_and_pad", "similarity")]
for framing, interp, blur, s in itertools.product(("crop_and_pad", "expand", "crop"), ("bilinear", "bicubic"), (0.0, 0.5), (9,)):
    ms, out = timed(lambda: ap.apply_motion(mk(), meta, (127, 127, 127), framing_mode=framing, interpolation=interp, motion_blur=blur,
                                            motion_blur_samples=s, ctx=ctx, keep_on_device=True))
    print(f"apply    {framing:13s} {interp:9s} blur={blur} S={s}: {ms:8.2f} ms")
    del out
ms, _ = timed(lambda: fp._stabilize_frames(mk(), "crop_and_pad", "similarity", True, 0.7, 0.5, 0.6, (127, 127, 127), 16.0, ctx=ctx, keep_on_device=True))
print(f"flow camera_lock: {ms:.2f} ms")
