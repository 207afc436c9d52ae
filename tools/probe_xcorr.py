"""GPU probe: device-resident clxcorrelate_fft_vcf throughput per FFT size (4 time-series inputs)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import __graft_entry__ as e
pkg = e.load_package()
def timeit(fn, iters=20):
    for _ in range(3): fn()
    s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    t.record(); torch.cuda.synchronize()
    return s.elapsed_time(t) / iters * 1e-3
sizes = [int(a) for a in sys.argv[1:]] or [16, 32, 64, 128, 256, 1024, 4096]
tot, nin = 1 << 24, 4
xs = [torch.randn(tot, 2, device="cuda") for _ in range(nin)]
ys = [torch.empty(tot, device="cuda") for _ in range(nin - 1)]
for itype in (2, 1):
    for n in sizes:
        blk = pkg.clxcorrelate_fft_vcf(n, nin, 1, 2, 0, 0, itype)
        dt = timeit(lambda: blk.work_device(tot // n, xs, ys))
        b = tot * (8 * nin + 4 * (nin - 1))
        print("xcorr type=%d N=%5d: %7.1f GS/s in  %.2f TB/s (%.1f%% of 8 TB/s)" % (itype, n, nin * tot / dt / 1e9, b / dt / 1e12, b / dt / 8e10))
