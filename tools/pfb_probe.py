import sys, os
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import __graft_entry__ as e
pkg = e.load_package(); o = e.load_oracle()
N = 1 << int(os.environ.get("PROBE_LOG2", "26"))
t = np.concatenate([o.firdes_low_pass(1.0, 64.0, 0.5, 0.0753), [0.0]]).astype(np.float32)
p = pkg.clPolyphaseChannelizer(1, 2, 0, 0, t, N, 64, 64, list(range(64)))
x = torch.randn(N + 2048 - 64, 2, device="cuda"); y = torch.empty(N, 2, device="cuda")
def ev(fn, it=20):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e-3 / it
dt = ev(lambda: p.work_device([x], [y]), int(os.environ.get("PROBE_IT", "20")))
print("pfb 64x32 N=2^%d: %.1f us %.1f GS/s %.3f of 8 TB/s" % (int(np.log2(N)), dt * 1e6, N / dt / 1e9, N * 16 / dt / 8e12))
