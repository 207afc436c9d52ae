"""clFFT rates for lengths that are not a power of two (tuning aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as e
pkg = e.load_package()
N = 1 << 25
a = torch.randn(N, 2, device="cuda"); c = torch.empty_like(a)
def ev(fn, it=5):
    for _ in range(2): fn()
    s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    t.record(); torch.cuda.synchronize()
    return s.elapsed_time(t) * 1e-3 / it
for n in [int(v) for v in sys.argv[1:]] or [1000, 1536, 2000, 3000, 4000, 5000, 6000, 10000, 12000, 15000, 20000, 50000, 100000]:
    blk = pkg.clFFT(n, pkg.CLFFT_FORWARD, np.blackman(n).astype(np.float32), pkg.DTYPE_COMPLEX, 1, 2, 0, 0, 0, 1, True)
    nv = N // n
    dt = ev(lambda: blk.work_device(nv, [a], [c]))
    print("clFFT %6d: %8.1f us  %6.1f GS/s  %.3f of 8 TB/s" % (n, dt * 1e6, nv * n / dt / 1e9, nv * n * 16 / dt / 8e12), flush=True)
