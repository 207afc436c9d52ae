#!/usr/bin/env python3
"""clFFT at small lengths that are not a power of two (the last pass' store runs are shorter than 64 bytes): us per 2^26 samples, GS/s.
MI355_FFT_MR_NO_COPY_OUT=1: the last pass stores for itself."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import __graft_entry__ as e
pkg = e.load_package()
ARGS = (1, 2, 0, 0)
tot = 1 << 26
x = torch.randn(tot, 2, device="cuda"); y = torch.empty_like(x)
for n in [int(v) for v in (sys.argv[1:] or ["12", "24", "48", "96", "80", "112", "20", "120", "1000"])]:
    nv = tot // n
    blk = pkg.clFFT(n, pkg.CLFFT_FORWARD, np.ones(n, np.float32), pkg.DTYPE_COMPLEX, *ARGS, 0, 1, True)
    for _ in range(3): blk.work_device(nv, [x], [y])
    torch.cuda.synchronize()
    s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): blk.work_device(nv, [x], [y])
    t.record(); torch.cuda.synchronize()
    us = s.elapsed_time(t) * 100
    print("n=%d us=%.1f hbm_frac=%.3f" % (n, us, nv * n * 16 / (us * 1e-6) / 8e12), flush=True)
