import torch, time
n,h,w=256,1080,1920
a=torch.rand((n,h,w,3),device='cuda'); b=torch.empty_like(a); m=torch.empty((n,h,w),device='cuda')
for _ in range(3): b.copy_(a); m.fill_(1.0)
torch.cuda.synchronize()
s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
ts=[]
for _ in range(8):
    s.record(); b.copy_(a); e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e))
ms=sorted(ts)[len(ts)//2]; print(f"torch copy 6.37GB->6.37GB: {ms:.3f} ms = {2*a.numel()*4/ms/1e6:.0f} GB/s")
ts=[]
for _ in range(8):
    s.record(); b.copy_(a); m.fill_(0.5); e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e))
ms=sorted(ts)[len(ts)//2]; print(f"copy + mask fill (14.86 GB): {ms:.3f} ms = {(2*a.numel()*4+m.numel()*4)/ms/1e6:.0f} GB/s")
