"""Rows > 64 (corner turn + persistent correlator): two independent integrations on two streams against one stream -- does the corner turn of one
overlap the correlation of the other (what an overlapped-slab form of one integration would rely on)?  Round 5 tuning aid."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as e
pkg = e.load_package()
N, F, T = int(os.environ.get("PROBE_N", "256")), int(os.environ.get("PROBE_F", "512")), 1024
it = 20
xe = [pkg.clXEngine(1, 2, 0, 0, False, pkg.DTYPE_BYTE, 1, N, 1, 0, F, T, []) for _ in range(2)]
xs = [torch.randint(-127, 128, (T, N, F, 1, 2), dtype=torch.int8, device="cuda") for _ in range(4)]
outs = [torch.zeros(xe[0].get_output_buffer_size(), 2, device="cuda") for _ in range(2)]
st = [torch.cuda.Stream(), torch.cuda.Stream()]
def run(two):
    for k in range(4):
        s = st[k & 1] if two else st[0]
        with torch.cuda.stream(s):
            xe[k & 1].xcorrelate_device(xs[k], outs[k & 1])
for two in (0, 1, 0, 1):
    for _ in range(2): run(two)
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True)
    a.record(torch.cuda.current_stream())
    for s in st: s.wait_event(a)
    for _ in range(it): run(two)
    ends = [torch.cuda.Event(enable_timing=True) for _ in st]
    for s, ev in zip(st, ends): ev.record(s)
    torch.cuda.synchronize()
    us = max(a.elapsed_time(ev) for ev in ends) * 1e3 / (it * 4)
    print("N=%d F=%d %s: %.1f us per integration" % (N, F, "two streams" if two else "one stream", us), flush=True)
