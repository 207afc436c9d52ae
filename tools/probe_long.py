"""Long filters (> 2048 taps): partitioned fast convolution vs the direct form, device resident."""
import os, sys
sys.path.insert(0, "/root/repo")
import torch, numpy as np
import __graft_entry__ as e
pkg = e.load_package()
def timeit(fn, iters=4):
    for _ in range(2): fn()
    s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    t.record(); torch.cuda.synchronize()
    return s.elapsed_time(t) / iters * 1e-3
n = 1 << int(os.environ.get("PROBE_LOG2", "25"))
x = torch.randn(n + 20000, 2, device="cuda"); y = torch.empty(n, 2, device="cuda")
rng = np.random.default_rng(0)
for ntaps in (2048, 2049, 3000, 4096, 6000, 8192, 16384):
    taps = rng.standard_normal(ntaps).astype(np.float32)
    r = []
    for use_time in (False, True):
        blk = pkg.clFilter(1, 2, 0, 0, 1, taps, 1, 0, use_time)
        dt = timeit(lambda: blk.work_device(n - ntaps, [x], [y]))
        r.append((n - ntaps) / dt / 1e9)
    print("ntaps=%5d: FFT mode %7.1f GS/s   direct form %6.1f GS/s" % (ntaps, r[0], r[1]))
