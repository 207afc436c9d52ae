"""GPU probe: device-resident clMathOp / clMathConst bandwidth (not part of the tests)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as e
pkg = e.load_package()
n = 1 << 26  # 64 Mi complex = 512 MiB per buffer
a = torch.randn(n, 2, device="cuda"); b = torch.randn(n, 2, device="cuda"); c = torch.empty_like(a)
def timeit(fn, iters=20):
    for _ in range(3): fn()
    s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    t.record(); torch.cuda.synchronize()
    return s.elapsed_time(t) / iters * 1e-3
mul = pkg.clMathOp(1, 1, 2, 0, 0, pkg.MATHOP_MULTIPLY)
dt = timeit(lambda: mul.work_device(n, [a, b], [c]))
print("clMathOp cmul : %.1f GS/s  %.2f TB/s" % (n / dt / 1e9, n * 24 / dt / 1e12))
mc = pkg.clMathConst(1, 1, 2, 0, 0, 2.0, pkg.MATHOP_MULTIPLY)
dt = timeit(lambda: mc.work_device(n, [a], [c]))
print("clMathConst   : %.1f GS/s  %.2f TB/s" % (n / dt / 1e9, n * 16 / dt / 1e12))
dt = timeit(lambda: c.copy_(a))
print("torch copy    : %.2f TB/s" % (n * 16 / dt / 1e12))
import numpy as np
for m in (8192, 1 << 20):
    ha = np.ones(m, np.complex64); hc = np.empty_like(ha)
    import time
    mul.work(m, [ha, ha], [hc])
    t0 = time.perf_counter()
    for _ in range(50): mul.work(m, [ha, ha], [hc])
    dt = (time.perf_counter() - t0) / 50
    print("host path n=%d: %.1f us/call, %.1f MS/s" % (m, dt * 1e6, m / dt / 1e6))
