"""X-engine config 5 with one input buffer (cache-warm) vs rotating over 8 input buffers (1 GiB: every integration comes from HBM)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as e
pkg = e.load_package()
N, F, T = 64, 1024, 1024
xs = [torch.randint(-127, 128, (T, N, F, 1, 2), dtype=torch.int8, device="cuda") for _ in range(8)]
blk = pkg.clXEngine(1, 2, 0, 0, False, pkg.DTYPE_BYTE, 1, N, 1, 0, F, T, [])
out = torch.zeros(blk.get_output_buffer_size(), 2, device="cuda")
def timeit(nbuf, iters=40):
    for i in range(4): blk.xcorrelate_device(xs[i % nbuf], out)
    s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(iters): blk.xcorrelate_device(xs[i % nbuf], out)
    t.record(); torch.cuda.synchronize()
    return s.elapsed_time(t) / iters * 1e3
for nbuf in (1, 2, 8):
    print("int8 config 5, %d input buffer(s) in rotation: %.1f us per integration" % (nbuf, timeit(nbuf)))
xc = [torch.randn(T, N, F, 1, 2, device="cuda") for _ in range(3)]
bc = pkg.clXEngine(1, 2, 0, 0, False, pkg.DTYPE_COMPLEX, 1, N, 1, 0, F, T, [])
oc = torch.zeros(bc.get_output_buffer_size(), 2, device="cuda")
def timeit_c(nbuf, iters=20):
    for i in range(3): bc.xcorrelate_device(xc[i % nbuf], oc)
    s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(iters): bc.xcorrelate_device(xc[i % nbuf], oc)
    t.record(); torch.cuda.synchronize()
    return s.elapsed_time(t) / iters * 1e3
for nbuf in (1, 3):
    print("cf32 config 5, %d input buffer(s) in rotation: %.1f us per integration" % (nbuf, timeit_c(nbuf)))
