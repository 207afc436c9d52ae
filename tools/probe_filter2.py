"""GPU probe: clFilter throughput across tap counts, modes and decimations (device resident, input samples/s)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
n = 1 << int(os.environ.get("PROBE_LOG2", "25"))
x = torch.randn(n + 4096, 2, device="cuda"); y = torch.empty(n, 2, device="cuda")
rng = np.random.default_rng(0)
for ntaps in (3, 17, 65, 129, 300, 1000, 2000):
    taps = rng.standard_normal(ntaps).astype(np.float32)
    for use_time in (False, True):
        for decim in (1, 4):
            blk = pkg.clFilter(1, 2, 0, 0, decim, taps, 1, 0, use_time)
            nout = (n - ntaps) // decim
            dt = timeit(lambda: blk.work_device(nout, [x], [y]))
            print("ntaps=%4d %s decim=%d: %7.1f GS/s in (%5.1f%% of 8 TB/s at 16 B/sample)" % (ntaps, "TD " if use_time else "FFT", decim, nout * decim / dt / 1e9, nout * decim * 16 / dt / 8e10))
