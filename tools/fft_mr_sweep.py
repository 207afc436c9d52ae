"""Mixed-radix clFFT: rate for every workgroup size x frames per iteration (tuning aid; MI355_FFT_MR_THREADS / _FRAMES are read at create)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as e
pkg = e.load_package()
N = 1 << 24
a = torch.randn(N, 2, device="cuda"); c = torch.empty_like(a)
def ev(fn, it=4):
    fn()
    s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    t.record(); torch.cuda.synchronize()
    return s.elapsed_time(t) * 1e-3 / it
for n in [int(v) for v in sys.argv[1:]]:
    res = []
    m = n; pt = 16
    for f in (7, 5, 3):
        while m % f == 0: m //= f; pt = min(pt, (16 // f) * f)
    for th in range(64, 1025, 64):
        fmax = th * pt // n
        for fr in sorted(set([1, 2, 3, 4, 6, 8, 12, 16, 24, 32, fmax])):
            if fr < 1 or fr > fmax: continue
            if (fr * n * 33 // 32 + 1) * 8 > 160 * 1024: continue
            os.environ["MI355_FFT_MR_THREADS"] = str(th); os.environ["MI355_FFT_MR_FRAMES"] = str(fr)
            blk = pkg.clFFT(n, pkg.CLFFT_FORWARD, np.blackman(n).astype(np.float32), pkg.DTYPE_COMPLEX, 1, 2, 0, 0, 0, 1, True)
            nv = N // n
            dt = ev(lambda: blk.work_device(nv, [a], [c]))
            res.append((nv * n / dt / 1e9, th, fr))
    res.sort(reverse=True)
    print(n, "best:", " ".join("%.0f(t%d,f%d)" % r for r in res[:6]), " worst: %.0f(t%d,f%d)" % res[-1], flush=True)
