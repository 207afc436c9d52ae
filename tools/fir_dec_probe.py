"""clFilter time-domain mode at decimations above 8 (tuning aid): GS/s of input by taps x decimation; MI355_FIR_MFMA=0 forces the LDS-staged kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as e
pkg = e.load_package()
N = 1 << 26
DECS = [int(v) for v in os.environ.get("PROBE_DECS", "10,16,32,50,100").split(",")]
a = torch.randn(N, 2, device="cuda"); c = torch.empty(N // min(DECS) + 1, 2, device="cuda")
rng = np.random.default_rng(1)
CT = bool(os.environ.get("PROBE_CTAPS"))  # complex taps (clComplexFilter)
for nt in (33, 65, 200, 400):
    t = (rng.standard_normal(nt) / np.sqrt(nt)).astype(np.float32)
    row = []
    for dec in DECS:
        blk = pkg.clFilter(1, 2, 0, 0, dec, t, 1, 0, True) if not CT else pkg.clComplexFilter(1, 2, 0, 0, dec, (t + 1j * t[::-1]).astype(np.complex64), 1, 0, use_time=True)
        nout = (N - nt) // dec
        fn = lambda: blk.work_device(nout, [a], [c])
        for _ in range(3): fn()
        s, f = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20): fn()
        f.record(); torch.cuda.synchronize()
        dt = s.elapsed_time(f) * 1e-3 / 20
        row.append("D=%3d %6.1f" % (dec, nout * dec / dt / 1e9))
    print("taps %3d: %s  GS/s of input" % (nt, "  ".join(row)), flush=True)
