#!/usr/bin/env python3
"""k_pfb_fir (streaming branch filters of the two-kernel channelizer path) against k_pfb_branches_t on the same handle: bit for bit; and timing."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import __graft_entry__ as e
pkg = e.load_package()
ARGS = (1, 2, 0, 0)
rng = np.random.default_rng(5)
bad = 0
for M, P, nst in [(1024, 32, 700), (1000, 7, 333), (48, 64, 1000), (64, 40, 5000), (600, 33, 100), (512, 64, 300), (2048, 4, 100), (7, 12, 5000), (100, 50, 777), (1024, 16, 9)]:
    K = M * P - (M // 3 if P % 2 else 0)
    taps = rng.standard_normal(K).astype(np.float32)
    blk = pkg.clPolyphaseChannelizer(*ARGS, taps, nst * M, M, M, list(range(M)))
    x = torch.randn(blk.ninput(), 2, device="cuda")
    y1 = torch.full((blk.noutput(), 2), 7.0, device="cuda"); y2 = torch.full((blk.noutput(), 2), 9.0, device="cuda")
    blk.work_device([x], [y1]); torch.cuda.synchronize()
    os.environ["MI355_PFB_NO_FIR_RING"] = "1"
    blk.work_device([x], [y2]); torch.cuda.synchronize()
    del os.environ["MI355_PFB_NO_FIR_RING"]
    same = torch.equal(y1, y2)
    print(M, P, nst, "same" if same else "DIFF %g" % (y1 - y2).abs().max().item(), flush=True)
    bad += not same
print("BAD", bad)
