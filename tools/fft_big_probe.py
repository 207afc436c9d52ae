"""clFFT rates for the multi-pass sizes (tuning aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as e
pkg = e.load_package()
N = 1 << int(os.environ.get("FFT_PROBE_LOG2", "26"))
a = torch.randn(N, 2, device="cuda"); c = torch.empty_like(a)
def ev(fn, it=10):
    for _ in range(2): fn()
    s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    t.record(); torch.cuda.synchronize()
    return s.elapsed_time(t) * 1e-3 / it
for lg in [int(v) for v in sys.argv[1:]] or [15, 16, 17, 20, 22, 24]:
    n = 1 << lg
    blk = pkg.clFFT(n, pkg.CLFFT_FORWARD, np.blackman(n).astype(np.float32), pkg.DTYPE_COMPLEX, 1, 2, 0, 0, 0, 1, True)
    nv = N // n
    dt = ev(lambda: blk.work_device(nv, [a], [c]))
    print("clFFT 2^%d: %7.1f us  %6.1f GS/s  %.3f of 8 TB/s" % (lg, dt * 1e6, N / dt / 1e9, N * 16 / dt / 8e12), flush=True)
