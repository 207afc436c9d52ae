#!/usr/bin/env python3
"""clPolyphaseChannelizer outside the specialised kernels: 100 / 20 / 10 / 200 / 1000 channels x 32 taps per arm, the bench row's call size.
Prints us per launch; under rocprofv3 --kernel-trace --stats the per-kernel split (branch filters / transform / map)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import __graft_entry__ as e
pkg = e.load_package()
ARGS = (1, 2, 0, 0)
n = 1 << 26
a = torch.randn(n, 2, device="cuda"); c = torch.empty_like(a)
rng = np.random.default_rng(1)
for M in [int(s) for s in (sys.argv[1:] or ["100", "20", "10", "200", "1000"])]:
    taps = rng.standard_normal(M * 32).astype(np.float32)
    buf = ((n // 2) // M) * M
    blk = pkg.clPolyphaseChannelizer(*ARGS, taps, buf, M, M, list(range(M)))
    xi, yo = a[:blk.ninput()], c[:blk.noutput()]
    for _ in range(5): blk.work_device([xi], [yo])
    torch.cuda.synchronize()
    s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): blk.work_device([xi], [yo])
    t.record(); torch.cuda.synchronize()
    us = s.elapsed_time(t) * 1e3 / 20
    print("M=%d items=%d us=%.1f hbm_frac=%.3f" % (M, buf, us, buf * 16 / (us * 1e-6) / 8e12), flush=True)
    del blk
