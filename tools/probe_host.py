"""Host-pointer work() throughput for large calls (pageable numpy buffers -> pinned staging -> device and back)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as e
pkg = e.load_package()
rng = np.random.default_rng(0)
def rate(fn, n, iters=5):
    fn(); t0 = time.perf_counter()
    for _ in range(iters): fn()
    return n * iters / (time.perf_counter() - t0) / 1e9
for logn in (20, 24, 26):
    n = 1 << logn
    x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64); y = np.empty_like(x)
    fft = pkg.clFFT(4096, pkg.CLFFT_FORWARD, np.blackman(4096).astype(np.float32), pkg.DTYPE_COMPLEX, 1, 2, 0, 0, 0, 1, True)
    mul = pkg.clMathConst(1, 1, 2, 0, 0, 2.0, pkg.MATHOP_MULTIPLY)
    taps = rng.standard_normal(65).astype(np.float32)
    xf = np.concatenate([np.zeros(64, np.complex64), x]); flt = pkg.clFilter(1, 2, 0, 0, 1, taps)
    print("n=2^%d: clFFT %.2f GS/s  clMathConst %.2f GS/s  clFilter %.2f GS/s   (8 B in + 8 B out per sample over PCIe)" % (
        logn, rate(lambda: fft.work(n // 4096, [x], [y]), n), rate(lambda: mul.work(n, [x], [y]), n), rate(lambda: flt.work(n, [xf], [y]), n)))
