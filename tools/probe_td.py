"""Direct-form (time-domain) FIR rates: real and complex taps, a few tap counts and decimations, device resident."""
import os, sys
sys.path.insert(0, "/root/repo")
import torch, numpy as np
import __graft_entry__ as e
pkg = e.load_package()
def timeit(fn, iters=8):
    for _ in range(2): fn()
    s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    t.record(); torch.cuda.synchronize()
    return s.elapsed_time(t) / iters * 1e-3
n = 1 << 26
x = torch.randn(n + 8192, 2, device="cuda"); y = torch.empty(n, 2, device="cuda")
rng = np.random.default_rng(0)
for ntaps in (9, 33, 65, 129, 300, 3000):
    for dec in (1, 2):
        taps = rng.standard_normal(ntaps).astype(np.float32)
        blk = pkg.clFilter(1, 2, 0, 0, dec, taps, 1, 0, True)  # time domain
        nout = (n - ntaps) // dec
        dt = timeit(lambda: blk.work_device(nout, [x], [y]), 4 if ntaps > 1000 else 8)
        ctaps = (taps + 1j * rng.standard_normal(ntaps)).astype(np.complex64)
        cb = pkg.clComplexFilter(1, 2, 0, 0, dec, ctaps, 1, 0, True)
        dtc = timeit(lambda: cb.work_device(nout, [x], [y]), 4 if ntaps > 1000 else 8)
        print("ntaps=%4d decim=%d: real taps %7.1f GS/s in   complex taps %7.1f GS/s in" % (ntaps, dec, nout * dec / dt / 1e9, nout * dec / dtc / 1e9))
