"""clXEngine config 5, several windows per launch, SUSTAINED (>= 80 ms of back-to-back launches per figure, inputs in rotation from HBM), switches of the
whole-line kernel A/B'd inside ONE process (the launch reads its environment per call).
usage: python tools/r05_lines_sustained.py "VAR=a,b,c" [nint ...]     e.g.  "MI355_XE_LINES_PF=0,2,3" 4 8 16 32   (tuning aid)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as e
pkg = e.load_package()
var, vals = sys.argv[1].split("=")
vals = vals.split(",")
nints = [int(x) for x in sys.argv[2:]] or [4, 8, 16, 32]
N, F, T = 64, int(os.environ.get("PROBE_F", "1024")), 1024
xe = pkg.clXEngine(1, 2, 0, 0, False, pkg.DTYPE_BYTE, 1, N, 1, 0, F, T, [])
for nint in nints:
    nbuf = max(2, -(-640_000_000 // (nint * T * N * F * 2)) + 1)
    xs = [torch.randint(-127, 128, (nint, T, N, F, 1, 2), dtype=torch.int8, device="cuda") for _ in range(nbuf)]
    out = torch.zeros(nint * xe.get_output_buffer_size(), 2, device="cuda")
    res = []
    for rep in range(2):
        for v in vals:
            os.environ[var] = v
            if var in ("MI355_XE_NO_LINES", "MI355_XE_NO_SPLIT") and v == "0": os.environ.pop(var)
            for k in range(4): xe.xcorrelate_n_device(nint, xs[k % nbuf], out)
            torch.cuda.synchronize()
            n = max(20, int(0.08 / (nint * 40e-6)))
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for k in range(n): xe.xcorrelate_n_device(nint, xs[k % nbuf], out)
            b.record(); torch.cuda.synchronize()
            res.append((v, a.elapsed_time(b) * 1e3 / n / nint))
    print("%2d windows per launch: " % nint + "  ".join("%s=%s %.2f" % (var.replace("MI355_XE_", ""), v, t) for v, t in res) + "  us per window")
    del xs, out
    torch.cuda.empty_cache()
