"""GPU probe: X-engine in all three input modes at 64 antennas x 1024 channels x 1024 frames."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as e
pkg = e.load_package()
def timeit(fn, iters=5):
    fn()
    s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    t.record(); torch.cuda.synchronize()
    return s.elapsed_time(t) / iters * 1e-3
N, F, T = 64, 1024, 1024
for name, dtype, npol, mk in (("ichar p1", pkg.DTYPE_BYTE, 1, lambda: torch.randint(-127, 128, (T, N, F, 1, 2), dtype=torch.int8, device="cuda")),
                              ("ichar p2", pkg.DTYPE_BYTE, 2, lambda: torch.randint(-127, 128, (T, N, F, 2, 2), dtype=torch.int8, device="cuda")),
                              ("packed4 ", pkg.DTYPE_PACKEDXY, 2, lambda: torch.randint(0, 256, (T, N, F, 2), dtype=torch.uint8, device="cuda")),
                              ("cf32 p1 ", pkg.DTYPE_COMPLEX, 1, lambda: torch.randn(T, N, F, 1, 2, device="cuda"))):
    x = mk()
    blk = pkg.clXEngine(1, 2, 0, 0, False, dtype, npol, N, 1, 0, F, T, [])
    out = torch.zeros(blk.get_output_buffer_size(), 2, device="cuda")
    dt = timeit(lambda: blk.xcorrelate_device(x, out))
    flop = 8.0 * F * (N * (N + 1) // 2) * T * npol * npol
    print("xengine %s: %8.1f us  %7.1f TFLOP/s  in %.0f MB" % (name, dt * 1e6, flop / dt / 1e12, x.numel() * x.element_size() / 1e6))
    del x, out, blk
