#!/usr/bin/env python3
"""k_pfb_mr (branch filters + transform in one kernel) against the two-kernel form of the same handle: bit for bit."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import __graft_entry__ as e
pkg = e.load_package()
ARGS = (1, 2, 0, 0)
rng = np.random.default_rng(3)
bad = 0
for M, P, nst, part in [(10, 32, 5000, 0), (12, 3, 1024, 0), (20, 8, 4099, 0), (24, 9, 2048, 1), (100, 32, 3001, 0), (100, 17, 50000, 1), (200, 16, 1500, 0),
                        (48, 32, 1025, 0), (360, 5, 2000, 0), (500, 32, 1111, 0), (96, 32, 300000, 0), (100, 32, 7, 0), (30, 4, 64, 0)]:
    K = M * P - (M // 3 if P % 2 else 0)
    taps = rng.standard_normal(K).astype(np.float32)
    buf = nst * M
    cmap = list(range(M)) if not part else [int(v) for v in rng.permutation(M)[:max(16, M // 2)]]
    blk = pkg.clPolyphaseChannelizer(*ARGS, taps, buf, M, len(cmap), cmap)
    x = torch.randn(blk.ninput(), 2, device="cuda")
    y1 = torch.full((blk.noutput(), 2), 7.0, device="cuda"); y2 = torch.full((blk.noutput(), 2), 9.0, device="cuda")
    blk.work_device([x], [y1]); torch.cuda.synchronize()
    os.environ["MI355_PFB_NO_MR_FUSED"] = "1"
    blk.work_device([x], [y2]); torch.cuda.synchronize()
    del os.environ["MI355_PFB_NO_MR_FUSED"]
    same = torch.equal(y1, y2)
    print(M, P, nst, part, "same" if same else "DIFF %g" % (y1 - y2).abs().max().item(), flush=True)
    bad += not same
print("BAD", bad)
