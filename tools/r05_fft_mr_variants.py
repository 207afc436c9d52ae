"""Mixed-radix clFFT: the rule's factorisation (variant 0) against the alternative (variant 1), rates per length (round 5: is a checked-in
per-length table worth having now that the choice is no longer timed at run time?)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as e
pkg = e.load_package()
N = 1 << 24
a = torch.randn(N, 2, device="cuda"); c = torch.empty_like(a)
def ev(fn, it=20):
    for _ in range(3): fn()
    s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    t.record(); torch.cuda.synchronize()
    return s.elapsed_time(t) * 1e-3 / it
for n in [int(v) for v in sys.argv[1:]]:
    r = []
    for var in ("0", "1"):
        os.environ["MI355_FFT_MR_VARIANT"] = var
        blk = pkg.clFFT(n, pkg.CLFFT_FORWARD, np.blackman(n).astype(np.float32), pkg.DTYPE_COMPLEX, 1, 2, 0, 0, 0, 1, True)
        nv = N // n
        r.append(nv * n / ev(lambda: blk.work_device(nv, [a], [c])) / 1e9)
        del blk
    print("%6d  rule %.0f GS/s  alternative %.0f GS/s  ratio %.2f" % (n, r[0], r[1], r[1] / r[0]), flush=True)
