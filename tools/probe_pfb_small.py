"""GPU probe: clPolyphaseChannelizer at the reference's call sizes (device resident), us per launch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as e
pkg = e.load_package(); o = e.load_oracle()
t2048 = np.concatenate([o.firdes_low_pass(1.0, 64.0, 0.5, 0.0753), [0.0]]).astype(np.float32)
def timeit(fn, iters=400):
    for _ in range(20): fn()
    s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    t.record(); torch.cuda.synchronize()
    return s.elapsed_time(t) / iters * 1e3
for M, tpa in ((64, 32), (128, 16), (256, 8), (64, 8)):
    taps = t2048[:M * tpa].copy()
    for buf in (8192, 65536, 262144, 1 << 20):
        if buf % M: continue
        p = pkg.clPolyphaseChannelizer(1, 2, 0, 0, taps, buf, M, M, list(range(M)))
        x = torch.randn(p.ninput(), 2, device="cuda"); y = torch.empty(p.noutput(), 2, device="cuda")
        us = timeit(lambda: p.work_device([x], [y]))
        print("M=%3d taps/arm=%2d buf=%7d: %6.2f us/launch %8.1f MS/s" % (M, tpa, buf, us, buf / us))
