"""GPU probe: host-pointer work() latency for scheduler-sized buffers (BASELINE configs[0]: 8192 complex items)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as e
pkg = e.load_package()
rng = np.random.default_rng(0)
def lat(fn, iters=300):
    for _ in range(20): fn()
    t0 = time.perf_counter()
    for _ in range(iters): fn()
    return (time.perf_counter() - t0) / iters * 1e6
for n in (1024, 8192, 32768, 65536, 262144):
    a = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64); b = a.copy(); c = np.empty_like(a)
    mul = pkg.clMathOp(pkg.DTYPE_COMPLEX, 1, 2, 0, 0, pkg.MATHOP_MULTIPLY)
    t = lat(lambda: mul.work(n, [a, b], [c]))
    assert np.allclose(c, a * b, rtol=1e-6)
    print("clMathOp.work %7d items: %7.1f us/call  %8.1f MS/s" % (n, t, n / t))
o = e.load_oracle()
taps = o.firdes_low_pass(1.0, 10e6, 1e6, 372000.0)
for n in (8192, 32768, 65536):
    x = (rng.standard_normal(n + 64) + 1j * rng.standard_normal(n + 64)).astype(np.complex64); y = np.empty(n, np.complex64)
    f = pkg.clFilter(1, 2, 0, 0, 1, taps, 1, 0, False)
    t = lat(lambda: f.work(n, [x], [y]))
    print("clFilter(fft,65).work %6d items: %7.1f us/call  %8.1f MS/s" % (n, t, n / t))
for nvec in (1, 2, 8):
    x = (rng.standard_normal(nvec * 4096) + 1j * rng.standard_normal(nvec * 4096)).astype(np.complex64); y = np.empty_like(x)
    f = pkg.clFFT(4096, pkg.CLFFT_FORWARD, np.blackman(4096).astype(np.float32), pkg.DTYPE_COMPLEX, 1, 2, 0, 0, 0, 1, True)
    t = lat(lambda: f.work(nvec, [x], [y]))
    print("clFFT(4096).work %d vectors: %7.1f us/call  %8.1f MS/s" % (nvec, t, nvec * 4096 / t))
t2048 = np.concatenate([o.firdes_low_pass(1.0, 64.0, 0.5, 0.0753), [0.0]]).astype(np.float32)
for buf in (8192, 65536):
    p = pkg.clPolyphaseChannelizer(1, 2, 0, 0, t2048, buf, 64, 64, list(range(64)))
    x = (rng.standard_normal(p.ninput()) + 1j * rng.standard_normal(p.ninput())).astype(np.complex64); y = np.empty(p.noutput(), np.complex64)
    t = lat(lambda: p.general_work(buf, None, [x], [y]))
    print("clPolyphaseChannelizer.work buf_items=%6d: %7.1f us/call  %8.1f MS/s" % (buf, t, buf / t))
