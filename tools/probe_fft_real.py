"""GPU probe: clFFT with real input (DTYPE_FLOAT): 4 B in + 8 B out per sample."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import __graft_entry__ as e
pkg = e.load_package()
def timeit(fn, iters=20):
    for _ in range(3): fn()
    s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    t.record(); torch.cuda.synchronize()
    return s.elapsed_time(t) / iters * 1e-3
tot = 1 << 26
x = torch.randn(tot, device="cuda"); y = torch.empty(tot, 2, device="cuda")
for n in [int(a) for a in sys.argv[1:]] or [256, 1024, 4096, 8192]:
    w = np.blackman(n).astype(np.float32)
    for d, name in ((pkg.CLFFT_FORWARD, "fwd"),):
        blk = pkg.clFFT(n, d, w, pkg.DTYPE_FLOAT, 1, 2, 0, 0, 0, 1, True)
        dt = timeit(lambda: blk.work_device(tot // n, [x], [y]))
        print("real-input fft N=%5d %s: %7.1f GS/s  %.2f TB/s (%.1f%% of 8 TB/s at 12 B/sample)" % (n, name, tot / dt / 1e9, tot * 12 / dt / 1e12, tot * 12 / dt / 8e10))
