"""GPU probe: interleaved A/B of headline clFFT launch variants inside ONE process (clock drift between processes is +-4 %)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as e
pkg = e.load_package()
N = int(os.environ.get("PROBE_N", "4096")); FR = (1 << 26) // N
w = np.blackman(N).astype(np.float32)
x = torch.randn(FR * N, 2, device="cuda"); y = torch.empty_like(x)
blk = pkg.clFFT(N, pkg.CLFFT_FORWARD, w, pkg.DTYPE_COMPLEX, 1, 2, 0, 0, 0, 1, True)
f = lambda: blk.work_device(FR, [x], [y])
variants = json.loads(sys.argv[1]) if len(sys.argv) > 1 else [{}, {"MI355_FFT_WG_PER_CU": "24"}, {"MI355_FFT_PREFETCH": "1", "MI355_FFT_WG_PER_CU": "2"}]
keys = sorted({k for v in variants for k in v})
for _ in range(300): f()
res = [[] for _ in variants]
for rnd in range(8):
    for i, v in enumerate(variants):
        for k in keys:
            os.environ.pop(k, None)
        os.environ.update(v)
        for _ in range(20): f()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(100): f()
        b.record(); torch.cuda.synchronize()
        res[i].append(a.elapsed_time(b) * 10)
for v, r in zip(variants, res):
    r2 = sorted(r)
    print("%-70s median %.2f  min %.2f  rounds %s" % (json.dumps(v), r2[len(r2) // 2], r2[0], " ".join("%.1f" % t for t in r)))
