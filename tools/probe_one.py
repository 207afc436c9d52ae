"""Run ONE block's device path a few times (for rocprofv3 runs): probe_one.py {fft|ols|fir|pfb|xe} [iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import __graft_entry__ as e
pkg = e.load_package(); o = e.load_oracle()
what = sys.argv[1]; iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
n = 1 << 26
if what == "fft":
    x = torch.randn(n, 2, device="cuda"); y = torch.empty_like(x)
    blk = pkg.clFFT(4096, pkg.CLFFT_FORWARD, np.blackman(4096).astype(np.float32), 1, 1, 2, 0, 0, 0, 1, True)
    fn = lambda: blk.work_device(n // 4096, [x], [y])
elif what in ("ols", "fir"):
    x = torch.randn(n + 64, 2, device="cuda"); y = torch.empty(n, 2, device="cuda")
    blk = pkg.clFilter(1, 2, 0, 0, 1, o.firdes_low_pass(1.0, 10e6, 1e6, 372000.0), 1, 0, what == "fir")
    fn = lambda: blk.work_device(n, [x], [y])
elif what == "pfb":
    taps = np.concatenate([o.firdes_low_pass(1.0, 64.0, 0.5, 0.0753), [0.0]]).astype(np.float32)
    blk = pkg.clPolyphaseChannelizer(1, 2, 0, 0, taps, n, 64, 64, list(range(64)))
    x = torch.randn(blk.ninput(), 2, device="cuda"); y = torch.empty(blk.noutput(), 2, device="cuda")
    fn = lambda: blk.work_device([x], [y])
else:
    N, F, T = 64, 1024, 1024
    x = torch.randint(-127, 128, (T, N, F, 1, 2), dtype=torch.int8, device="cuda")
    blk = pkg.clXEngine(1, 2, 0, 0, False, pkg.DTYPE_BYTE, 1, N, 1, 0, F, T, [])
    y = torch.zeros(blk.get_output_buffer_size(), 2, device="cuda")
    fn = lambda: blk.xcorrelate_device(x, y)
for _ in range(iters): fn()
torch.cuda.synchronize()
