"""GPU probe: remaining elementwise blocks, device resident."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import __graft_entry__ as e
pkg = e.load_package()
def timeit(fn, iters=10):
    for _ in range(2): fn()
    s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    t.record(); torch.cuda.synchronize()
    return s.elapsed_time(t) / iters * 1e-3
n = 1 << 26
c = torch.randn(n + 1, 2, device="cuda"); f1 = torch.rand(n, device="cuda") + 0.5; f2 = torch.rand(n, device="cuda") + 0.5
of = torch.empty(n, device="cuda"); of2 = torch.empty(n, device="cuda"); oc = torch.empty(n, 2, device="cuda")
A = (1, 2, 0, 0)
cases = [("clLog", pkg.clLog(*A, 10.0, 0.0), [f1], [of], 8), ("clSNR", pkg.clSNR(*A, 10.0, 0.0), [f1, f2], [of], 12),
         ("clComplexToMag", pkg.clComplexToMag(*A), [c], [of], 12), ("clComplexToArg", pkg.clComplexToArg(*A), [c], [of], 12),
         ("clComplexToMagPhase", pkg.clComplexToMagPhase(*A), [c], [of, of2], 16), ("clMagPhaseToComplex", pkg.clMagPhaseToComplex(*A), [f1, f2], [oc], 16),
         ("clQuadratureDemod", pkg.clQuadratureDemod(1.0, *A), [c], [of], 12)]
for name, blk, i, o, bps in cases:
    dt = timeit(lambda: blk.work_device(n, i, o))
    print("%-22s %7.1f GS/s  %.2f TB/s (%.1f%% of 8 TB/s)" % (name, n / dt / 1e9, n * bps / dt / 1e12, n * bps / dt / 8e10))
