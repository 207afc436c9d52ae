"""k_fir_dec2: GS/s of input by decimation and LDS padding shift (MI355_FIR_DEC2_PAD; 31 = none), 65 real taps.  usage: python tools/r06_dec2_pad_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as e
pkg = e.load_package()
N = 1 << 26
a = torch.randn(N, 2, device="cuda"); c = torch.empty(N // 6 + 1, 2, device="cuda")
rng = np.random.default_rng(1)
os.environ["MI355_FIR_DEC_KERNEL"] = "lds"
nt = int(os.environ.get("PROBE_TAPS", "65"))
t = (rng.standard_normal(nt) / np.sqrt(nt)).astype(np.float32)
for dec in list(range(6, 41)) + [48, 50, 64, 100]:
    row = []
    for sh in ("2", "3", "4", "5", "6", "31"):
        os.environ["MI355_FIR_DEC2_PAD"] = sh
        blk = pkg.clFilter(1, 2, 0, 0, dec, t, 1, 0, True)
        nout = (N - nt) // dec
        fn = lambda: blk.work_device(nout, [a], [c])
        for _ in range(3): fn()
        s, f = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10): fn()
        f.record(); torch.cuda.synchronize()
        row.append(nout * dec / (s.elapsed_time(f) * 1e-3 / 10) / 1e9)
    best = max(range(len(row)), key=lambda i: row[i])
    print("D=%3d  pad shift 2/3/4/5/6/none: %s   best %s" % (dec, " ".join("%6.1f" % v for v in row), ("2", "3", "4", "5", "6", "none")[best]), flush=True)
