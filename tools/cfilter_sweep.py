"""clComplexFilter (complex taps) over tap counts x decimations x both modes (tuning aid: looks for rate cliffs)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as e
pkg = e.load_package()
n = 1 << 26
a = torch.randn(n, 2, device="cuda"); c = torch.empty_like(a)
def ev(fn, it=5):
    for _ in range(2): fn()
    s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    t.record(); torch.cuda.synchronize()
    return s.elapsed_time(t) * 1e-3 / it
rng = np.random.default_rng(0)
for ntaps in (5, 33, 65, 200, 600, 3000):
    taps = ((rng.standard_normal(ntaps) + 1j * rng.standard_normal(ntaps)) / ntaps).astype(np.complex64)
    for dec in (1, 3, 8, 10, 16, 50):
        for use_time in (False, True):
            try:
                blk = pkg.clComplexFilter(1, 2, 0, 0, dec, taps, 1, 0, use_time=use_time)
                nout = (n - ntaps) // dec
                dt = ev(lambda: blk.work_device(nout, [a], [c]))
                print("ctaps %4d decim %3d %s: %8.1f us  input %6.1f GS/s" % (ntaps, dec, "time" if use_time else "fft ", dt * 1e6, nout * dec / dt / 1e9), flush=True)
            except Exception as ex:
                print("ctaps %d decim %d %s: %s" % (ntaps, dec, use_time, str(ex)[:80]), flush=True)
