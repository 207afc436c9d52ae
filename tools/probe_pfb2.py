"""GPU probe: PFB time vs taps per arm (which phase dominates?)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import __graft_entry__ as e
pkg = e.load_package()
def timeit(fn, iters=10):
    for _ in range(2): fn()
    s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    t.record(); torch.cuda.synchronize()
    return s.elapsed_time(t) / iters * 1e-3
buf = 1 << 26
for M in (8, 16, 32, 64, 128, 256):
    for per_arm in (8, 16, 32):
        taps = np.random.default_rng(1).standard_normal(M * per_arm).astype(np.float32)
        blk = pkg.clPolyphaseChannelizer(1, 2, 0, 0, taps, buf, M, M, list(range(M)))
        x = torch.randn(blk.ninput(), 2, device="cuda"); y = torch.empty(blk.noutput(), 2, device="cuda")
        dt = timeit(lambda: blk.work_device([x], [y]))
        print("pfb M=%3d taps/arm=%2d: %7.1f us %7.1f GS/s (%.1f%% of 8 TB/s)" % (M, per_arm, dt * 1e6, buf / dt / 1e9, buf * 16 / dt / 8e10))
        del x, y, blk
