"""clXEngine config 5, one window per call from HBM: calls on ONE stream against calls alternating between TWO streams (two handles, so two
partial-sum workspaces): what overlapping the tail of launch k with the start of launch k+1 is worth (round 5 tuning aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as e
pkg = e.load_package()
N, F, T = 64, 1024, 1024
nbuf, it = 4, int(os.environ.get("PROBE_IT", "40"))
xe = [pkg.clXEngine(1, 2, 0, 0, False, pkg.DTYPE_BYTE, 1, N, 1, 0, F, T, []) for _ in range(2)]
xs = [torch.randint(-127, 128, (T, N, F, 1, 2), dtype=torch.int8, device="cuda") for _ in range(nbuf)]
outs = [torch.zeros(xe[0].get_output_buffer_size(), 2, device="cuda") for _ in range(nbuf)]
st = [torch.cuda.Stream(), torch.cuda.Stream()]
def run(two):
    for k in range(nbuf):
        s = st[k & 1] if two else st[0]
        with torch.cuda.stream(s):
            xe[k & 1 if two else 0].xcorrelate_device(xs[k], outs[k])
for two in (0, 1, 0, 1):
    for _ in range(3): run(two)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    cur = torch.cuda.current_stream()
    a.record(cur)
    for s in st: s.wait_event(a)
    for _ in range(it): run(two)
    ends = [torch.cuda.Event(enable_timing=True) for _ in st]
    for s, ev in zip(st, ends): ev.record(s)
    torch.cuda.synchronize()
    us = max(a.elapsed_time(ev) for ev in ends) * 1e3 / (it * nbuf)
    print("%s: %.1f us per window" % ("two streams, two handles" if two else "one stream", us), flush=True)
ref = torch.zeros_like(outs[0]); xe[0].xcorrelate_device(xs[0], ref); torch.cuda.synchronize()
print("outputs of the overlapped runs equal a lone launch's:", torch.equal(ref, outs[0]))
