"""GPU probe: interleaved A/B of env-var variants for one block inside ONE process.
usage: probe_ab.py <block> '<json list of env dicts>'   block: filter65 | pfb | mathconst | fir65 | fft32768 | filter3000"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as e
pkg = e.load_package(); o = e.load_oracle()
which = sys.argv[1]
variants = json.loads(sys.argv[2])
args = (1, 2, 0, 0)
n = 1 << 26
a = torch.randn(n, 2, device="cuda"); c = torch.empty_like(a)
taps65 = o.firdes_low_pass(1.0, 10e6, 1e6, 372000.0)
if which == "filter65":
    blk = pkg.clFilter(*args, 1, taps65, 1, 0, False); f = lambda: blk.work_device(n - 64, [a], [c])
elif which == "fir65":
    blk = pkg.clFilter(*args, 1, taps65, 1, 0, True); f = lambda: blk.work_device(n - 64, [a], [c])
elif which == "filter3000":
    t = (np.random.default_rng(3000).standard_normal(3000) / 55).astype(np.float32)
    blk = pkg.clFilter(*args, 1, t, 1, 0, False); f = lambda: blk.work_device(n - 3000, [a], [c])
elif which == "pfb":
    t2048 = np.concatenate([o.firdes_low_pass(1.0, 64.0, 0.5, 0.0753), [0.0]]).astype(np.float32)
    buf = (1 << 26) - (1 << 16)
    blk = pkg.clPolyphaseChannelizer(*args, t2048, buf, 64, 64, list(range(64))); f = lambda: blk.work_device([a], [c])
elif which == "mathconst":
    blk = pkg.clMathConst(pkg.DTYPE_COMPLEX, *args, 2.0, pkg.MATHOP_MULTIPLY); f = lambda: blk.work_device(n, [a], [c])
elif which.startswith("fft"):
    N = int(which[3:]); blk = pkg.clFFT(N, pkg.CLFFT_FORWARD, np.blackman(N).astype(np.float32), pkg.DTYPE_COMPLEX, *args, 0, 1, True)
    f = lambda: blk.work_device(n // N, [a], [c])
keys = sorted({k for v in variants for k in v})
for _ in range(100): f()
res = [[] for _ in variants]
for rnd in range(6):
    for i, v in enumerate(variants):
        for k in keys: os.environ.pop(k, None)
        os.environ.update(v)
        for _ in range(10): f()
        s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(50): f()
        t.record(); torch.cuda.synchronize()
        res[i].append(s.elapsed_time(t) * 20)
for v, r in zip(variants, res):
    r2 = sorted(r)
    print("%-12s %-60s median %.2f  min %.2f" % (which, json.dumps(v), r2[len(r2) // 2], r2[0]))
