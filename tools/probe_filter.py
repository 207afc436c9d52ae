"""GPU probe: device-resident clFilter throughput (65 taps, decim 1)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import __graft_entry__ as e
pkg = e.load_package(); o = e.load_oracle()
def timeit(fn, iters=10):
    for _ in range(2): fn()
    s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    t.record(); torch.cuda.synchronize()
    return s.elapsed_time(t) / iters * 1e-3
taps = o.firdes_low_pass(1.0, 10e6, 1e6, 372000.0)
n = 1 << 26
x = torch.randn(n + 64, 2, device="cuda"); y = torch.empty(n, 2, device="cuda")
for nf in ([0] + [int(a) for a in sys.argv[1:]]):
    if nf: os.environ["MI355_FILTER_FFT"] = str(nf)
    blk = pkg.clFilter(1, 2, 0, 0, 1, taps, 1, 0, False)
    dt = timeit(lambda: blk.work_device(n, [x], [y]))
    print("fft-filter 65 taps NF=%4d: %7.1f GS/s  %.2f TB/s (%.1f%% of 8 TB/s)" % (blk.fftsize(), n / dt / 1e9, n * 16 / dt / 1e12, n * 16 / dt / 8e10))
blk = pkg.clFilter(1, 2, 0, 0, 1, taps, 1, 0, True)
dt = timeit(lambda: blk.work_device(n, [x], [y]))
print("td-fir     65 taps        : %7.1f GS/s  %.2f TB/s" % (n / dt / 1e9, n * 16 / dt / 1e12))
ct = (taps * np.exp(1j * np.pi * np.arange(65) / 8)).astype(np.complex64)
blk = pkg.clComplexFilter(1, 2, 0, 0, 1, ct, 1, 0, use_time=True)
dt = timeit(lambda: blk.work_device(n, [x], [y]))
print("td-fir ccc 65 taps        : %7.1f GS/s" % (n / dt / 1e9))
blk = pkg.clComplexFilter(1, 2, 0, 0, 1, ct, 1, 0, use_time=False)
dt = timeit(lambda: blk.work_device(n, [x], [y]))
print("fft-filter ccc 65 taps    : %7.1f GS/s" % (n / dt / 1e9))
