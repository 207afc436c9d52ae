"""clXEngine config 5 with the input rotating over NBUF distinct windows (NBUF x 134 MB): does the single-buffer figure owe anything to the 256 MiB
Infinity Cache?  usage: PROBE_NBUF=1|2|4|8 python tools/xe_rotate_probe.py   (tuning aid)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as e
pkg = e.load_package()
N, F, T, NPOL = int(os.environ.get("PROBE_N", "64")), int(os.environ.get("PROBE_F", "1024")), 1024, int(os.environ.get("PROBE_NPOL", "1"))
nint = int(os.environ.get("PROBE_NINT", "1"))
nbuf = int(os.environ.get("PROBE_NBUF", "4")); it = int(os.environ.get("PROBE_IT", "50"))
xe = pkg.clXEngine(1, 2, 0, 0, False, pkg.DTYPE_BYTE, NPOL, N, 1, 0, F, T, [])
xs = [torch.randint(-127, 128, (nint, T, N, F, NPOL, 2), dtype=torch.int8, device="cuda") for _ in range(nbuf)]
outs = [torch.zeros(nint * xe.get_output_buffer_size(), 2, device="cuda") for _ in range(nbuf)]
shift = int(os.environ.get("PROBE_SHIFT", "0"))  # > 0: a small kernel of `shift` workgroups in front of every launch (the dispatcher's round robin over the XCDs goes on from there)
junk = torch.zeros(shift * 256, device="cuda") if shift else None
def run():
    for k in range(nbuf):
        if shift: junk.add_(1.0)
        if nint > 1: xe.xcorrelate_n_device(nint, xs[k], outs[k])
        else: xe.xcorrelate_device(xs[k], outs[k])
fresh = int(os.environ.get("PROBE_FRESH", "0"))  # 1: every window is written by a device copy right before its launch (a producer kernel)
master = xs[0].clone() if fresh else None
for _ in range(3): run()
if fresh:
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(it * nbuf)]
    for i in range(it * nbuf):
        k = i % nbuf
        xs[k].copy_(master)
        ev[i][0].record()
        if nint > 1: xe.xcorrelate_n_device(nint, xs[k], outs[k])
        else: xe.xcorrelate_device(xs[k], outs[k])
        ev[i][1].record()
    torch.cuda.synchronize()
    us = sum(a.elapsed_time(b) for a, b in ev) * 1e3 / (it * nbuf)
else:
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): run()
    b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) * 1e3 / (it * nbuf)
print("N=%d F=%d npol=%d windows/launch=%d, %d distinct inputs in rotation%s (%s): %.1f us per launch" % (N, F, NPOL, nint, nbuf, ", each written right before its launch" if fresh else "", os.environ.get("MI355_XE_INKERNEL_REDUCE", "default"), us))
