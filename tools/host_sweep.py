"""Host-pointer work() calls over the call size (one 4096-point vector ... 2^24 samples), clFFT / clMathConst / clFilter: looks for
steps between the direct path (kernel on the pinned staging), the staged path and its chunking (tuning aid)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as e
pkg = e.load_package()
rng = np.random.default_rng(0)
fft = pkg.clFFT(4096, pkg.CLFFT_FORWARD, np.blackman(4096).astype(np.float32), pkg.DTYPE_COMPLEX, 1, 2, 0, 0, 0, 1, True)
mc = pkg.clMathConst(pkg.DTYPE_COMPLEX, 1, 2, 0, 0, 2.0, pkg.MATHOP_MULTIPLY)
taps = (rng.standard_normal(65) / 65).astype(np.float32)
fl = pkg.clFilter(1, 2, 0, 0, 1, taps)
x = (rng.standard_normal(1 << 24) + 1j * rng.standard_normal(1 << 24)).astype(np.complex64)
y = np.empty_like(x)
def t(fn, n):
    reps = max(3, min(200, int(2e7 // n)))
    fn(); fn()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    return (time.perf_counter() - t0) / reps
for lg in range(12, 25):
    n = 1 << lg
    a = t(lambda: fft.work(n // 4096, [x[:n]], [y[:n]]), n)
    b = t(lambda: mc.work(n, [x[:n]], [y[:n]]), n)
    c = t(lambda: fl.work(n - 64, [x[:n]], [y[:n - 64]]), n)
    print("2^%2d samples: clFFT %9.1f us %7.1f MS/s | clMathConst %9.1f us %7.1f MS/s | clFilter %9.1f us %7.1f MS/s" % (lg, a * 1e6, n / a / 1e6, b * 1e6, n / b / 1e6, c * 1e6, n / c / 1e6), flush=True)
