"""Time the complex-float X-engine (device path): probe_xe_cf32.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import __graft_entry__ as e
pkg = e.load_package()
for npol in (1, 2):
    N, F, T = 64, 1024 // npol, 1024
    x = torch.randn(T, N, F, npol, 2, device="cuda")
    blk = pkg.clXEngine(1, 2, 0, 0, False, pkg.DTYPE_COMPLEX, npol, N, 1, 0, F, T, [])
    y = torch.zeros(blk.get_output_buffer_size(), 2, device="cuda")
    for _ in range(3): blk.xcorrelate_device(x, y)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): blk.xcorrelate_device(x, y)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print(f"xengine cf32 npol={npol} N={N} F={F} T={T}: {dt*1e6:.1f} us  in={x.numel()*4/dt/1e9:.0f} GB/s")
