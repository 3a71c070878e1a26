import sys, time
sys.path.insert(0,'/root/repo')
import numpy as np, torch
import __graft_entry__ as graft
graft.load_package()
import bench
from vstab_amd import flow_pipeline as fp, host_math as hm, native, apply_pipeline as ap
ctx = native.Context(0); ctx.set_timing(True)
n,h,w=256,1080,1920
frames = bench.synth_clip(n, 0, h, w, torch.device("cuda",0))
mk = lambda: hm.VideoContext([None]*n, hm.FrameAdapter(np.dtype(np.float32), False, "0_1", "torch", False), w, h, 3, None, "sequence", {}, batch=frames)
for r in range(3):
    torch.cuda.synchronize(); t0=time.perf_counter()
    res = fp._stabilize_frames(mk(), "crop", "similarity", False, 0.7, 0.5, 0.6, (127,127,127), 16.0, ctx=ctx, keep_on_device=True)
    torch.cuda.synchronize(); t1=time.perf_counter()
    out = ap.apply_motion(mk(), res.meta, (127,127,127), framing_mode="crop", interpolation="bilinear", ctx=ctx, keep_on_device=True)
    torch.cuda.synchronize(); t2=time.perf_counter()
    print(f"crop: flow {1e3*(t1-t0):.1f} ms, status {res.meta['framing']['keep_fov_status']}, scale {res.meta['framing']['stabilization_scale']:.3f}; motion-apply crop {1e3*(t2-t1):.1f} ms, fallback {out.meta['motion_apply'].get('framing_fallback')}")
