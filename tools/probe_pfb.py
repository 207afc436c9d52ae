"""GPU probe: device-resident polyphase channelizer throughput (64 ch x 32 taps/arm)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import __graft_entry__ as e
pkg = e.load_package(); o = e.load_oracle()
def timeit(fn, iters=10):
    for _ in range(2): fn()
    s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    t.record(); torch.cuda.synchronize()
    return s.elapsed_time(t) / iters * 1e-3
taps = np.concatenate([o.firdes_low_pass(1.0, 64.0, 0.5, 0.0753), [0.0]]).astype(np.float32)
for buf in (65536, 1 << 20, 1 << 26):
    blk = pkg.clPolyphaseChannelizer(1, 2, 0, 0, taps, buf, 64, 64, list(range(64)))
    x = torch.randn(blk.ninput(), 2, device="cuda"); y = torch.empty(blk.noutput(), 2, device="cuda")
    dt = timeit(lambda: blk.work_device([x], [y]), 20)
    print("pfb 64ch x32 buf_items=%9d: %8.1f us  %7.1f GS/s  %.2f TB/s (%.1f%% of 8 TB/s)" % (buf, dt * 1e6, buf / dt / 1e9, buf * 16 / dt / 1e12, buf * 16 / dt / 8e10))
