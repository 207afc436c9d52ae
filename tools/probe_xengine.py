"""GPU probe: device-resident X-engine, 64 antennas x 1024 channels x 1024 frames, IChar."""
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
import sys
for (N, F, T, npol) in [(64, 1024, 1024, 1), (64, 1024, 1024, 2)][:int(sys.argv[1]) if len(sys.argv) > 1 else 2]:
    x = torch.randint(-127, 128, (T, N, F, npol, 2), dtype=torch.int8, device="cuda")
    blk = pkg.clXEngine(1, 2, 0, 0, False, pkg.DTYPE_BYTE, npol, N, 1, 0, F, T, [])
    out = torch.zeros(blk.get_output_buffer_size(), 2, device="cuda")
    dt = timeit(lambda: blk.xcorrelate_device(x, out))
    B = N * (N + 1) // 2
    flop = 8.0 * F * B * T * npol * npol
    alg = x.numel() + out.numel() * 4
    print("xengine N=%d F=%d T=%d npol=%d: %.1f us  %.1f TFLOP/s (%.1f%% of 5 POPS i8)  alg %.2f TB/s (%.1f%% HBM)  %.1f GS/s in" %
          (N, F, T, npol, dt * 1e6, flop / dt / 1e12, flop / dt / 5e13, alg / dt / 1e12, alg / dt / 8e10, N * npol * F * T / dt / 1e9))
