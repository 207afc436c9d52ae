"""X-engine host path (submit / wait double buffering) timing split (tuning aid)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as e
pkg = e.load_package()
N, F, T = 64, 1024, 1024
rng = np.random.default_rng(0)
xe = pkg.clXEngine(1, 2, 0, 0, False, pkg.DTYPE_BYTE, 1, N, 1, 0, F, T, [])
xi = rng.integers(-127, 128, size=T * N * F * 2, dtype=np.int64).astype(np.int8)
vis = np.empty(xe.get_output_buffer_size(), np.complex64)
xe.xcorrelate(xi, vis)
xe.xcorrelate(xi, vis)
k = 8
ts, tw = [], []
t0 = time.perf_counter()
xe.submit(xi)
for _ in range(k - 1):
    a = time.perf_counter(); xe.submit(xi); b = time.perf_counter(); xe.wait(vis); c = time.perf_counter()
    ts.append(b - a); tw.append(c - b)
xe.wait(vis)
dt = (time.perf_counter() - t0) / k
print("per integration %.2f ms; submit %.2f ms (min %.2f), wait %.2f ms (min %.2f)" % (dt * 1e3, np.mean(ts) * 1e3, min(ts) * 1e3, np.mean(tw) * 1e3, min(tw) * 1e3))
t0 = time.perf_counter()
for _ in range(3): xe.xcorrelate(xi, vis)
print("synchronous xcorrelate: %.2f ms" % ((time.perf_counter() - t0) / 3 * 1e3))
