import torch
def t(fn, it=20):
    for _ in range(3): fn()
    a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b)*1e-3/it
for mb in (64,135,512,2048):
    x=torch.empty(mb*1024*1024//4,dtype=torch.float32,device='cuda')
    dt=t(lambda: x.zero_())
    print("zero_ %d MB: %.1f us %.2f TB/s"%(mb,dt*1e6,x.numel()*4/dt/1e12))
    y=torch.empty_like(x)
    dt=t(lambda: y.copy_(x))
    print("copy %d MB: %.1f us %.2f TB/s (r+w)"%(mb,dt*1e6,2*x.numel()*4/dt/1e12))
