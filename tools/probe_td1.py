"""One direct-form FIR configuration in a loop (for counter passes): probe_td1.py <ntaps> [complex]"""
import sys
sys.path.insert(0, "/root/repo")
import torch, numpy as np
import __graft_entry__ as e
pkg = e.load_package()
ntaps = int(sys.argv[1]); cplx = len(sys.argv) > 2
n = 1 << 24
x = torch.randn(n + 8192, 2, device="cuda"); y = torch.empty(n, 2, device="cuda")
taps = np.random.default_rng(0).standard_normal(ntaps).astype(np.float32)
blk = pkg.clComplexFilter(1, 2, 0, 0, 1, taps.astype(np.complex64), 1, 0, True) if cplx else pkg.clFilter(1, 2, 0, 0, 1, taps, 1, 0, True)
for _ in range(4): blk.work_device(n - ntaps, [x], [y])
torch.cuda.synchronize()
