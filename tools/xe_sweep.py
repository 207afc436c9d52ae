"""clXEngine over unusual geometries (antenna counts that are not a multiple of 16, channel counts that are not whole 128-byte rows,
integrations that are not a multiple of 32, all three input types): a hunt for rate cliffs (tuning aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as e
pkg = e.load_package()
def ev(fn, it=5):
    for _ in range(2): fn()
    s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    t.record(); torch.cuda.synchronize()
    return s.elapsed_time(t) * 1e-3 / it
cases = [(64,1024,1024,1),(60,1024,1024,1),(50,1000,1000,1),(20,1000,1000,1),(10,256,1024,1),(64,1000,1024,1),(64,1024,1000,1),(64,100,1024,1),
         (32,1024,1024,2),(30,1000,1000,2),(12,500,500,2),(100,512,1024,1),(130,500,1000,1),(7,64,128,1),(2,4096,4096,1),(64,16,16384,1),(64,4096,256,1)]
for kind, name, esz in ((pkg.DTYPE_BYTE, "ichar", 2), (pkg.DTYPE_COMPLEX, "cf32", 8), (pkg.DTYPE_PACKEDXY, "packed4", 1)):
    for N, F, T, npol in cases:
        if kind == pkg.DTYPE_PACKEDXY and npol == 1: continue
        if kind == pkg.DTYPE_COMPLEX and N * F * T * npol * 8 > (6 << 30): continue
        try:
            xe = pkg.clXEngine(1, 2, 0, 0, False, kind, npol, N, 1, 0, F, T, [])
            nbytes = xe.input_bytes()
            x = torch.randint(-127, 128, (nbytes,), dtype=torch.int8, device="cuda") if kind != pkg.DTYPE_COMPLEX else torch.randn(nbytes // 4, device="cuda")
            out = torch.zeros(xe.get_output_buffer_size(), 2, device="cuda")
            dt = ev(lambda: xe.xcorrelate_device(x, out))
            print("XE %-7s N=%3d F=%4d T=%5d npol=%d: %8.1f us  input %6.2f TB/s  (%5.1f GS/s)" % (name, N, F, T, npol, dt * 1e6, nbytes / dt / 1e12, N * F * T * npol / dt / 1e9), flush=True)
        except Exception as ex:
            print("XE %s N=%d F=%d T=%d npol=%d: %s" % (name, N, F, T, npol, str(ex)[:120]), flush=True)
