"""GPU probe: device-resident clFFT throughput."""
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
sizes = [int(a) for a in sys.argv[1:]] or [4096]
tot = 1 << 26
x = torch.randn(tot, 2, device="cuda"); y = torch.empty_like(x)
for n in sizes:
    nvec = tot // n
    if n & (n - 1): nvec = min(nvec, (int(os.environ.get("PROBE_CZ_LOG2", "22")) and (1 << int(os.environ.get("PROBE_CZ_LOG2", "22")))) // n)  # chirp-z sizes: smaller batch
    w = np.blackman(n).astype(np.float32)
    blk = pkg.clFFT(n, pkg.CLFFT_FORWARD, w, pkg.DTYPE_COMPLEX, 1, 2, 0, 0, 0, 1, True)
    dt = timeit(lambda: blk.work_device(nvec, [x], [y]))
    cnt = nvec * n
    print("fft N=%5d win+shift: %7.1f GS/s  %.2f TB/s  (%.1f%% of 8 TB/s)" % (n, cnt / dt / 1e9, cnt * 16 / dt / 1e12, cnt * 16 / dt / 8e10))
