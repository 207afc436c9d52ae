"""GPU probe: the 65536-item channelizer call replayed from a HIP graph (device-side time per launch, no host launch cost)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as e
pkg = e.load_package(); o = e.load_oracle()
t2048 = np.concatenate([o.firdes_low_pass(1.0, 64.0, 0.5, 0.0753), [0.0]]).astype(np.float32)
buf = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
p = pkg.clPolyphaseChannelizer(1, 2, 0, 0, t2048, buf, 64, 64, list(range(64)))
x = torch.randn(p.ninput(), 2, device="cuda"); y = torch.empty(p.noutput(), 2, device="cuda")
p.work_device([x], [y]); torch.cuda.synchronize()
n = 256
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    with torch.cuda.graph(g, stream=s):
        for _ in range(n): p.work_device([x], [y])
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
g.replay(); torch.cuda.synchronize()
a.record()
for _ in range(10): g.replay()
b.record(); torch.cuda.synchronize()
print("buf=%d graph replay: %.2f us per launch" % (buf, a.elapsed_time(b) * 1e3 / (10 * n)))
