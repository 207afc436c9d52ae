import os, sys
sys.path.insert(0, "/root/repo")
import torch, numpy as np
import __graft_entry__ as e
pkg = e.load_package()
def timeit(fn, iters=8):
    for _ in range(2): fn()
    s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    t.record(); torch.cuda.synchronize()
    return s.elapsed_time(t) / iters * 1e-3
n = 1 << int(os.environ.get("PROBE_LOG2", "25"))
x = torch.randn(n + 4096, 2, device="cuda"); y = torch.empty(n, 2, device="cuda")
rng = np.random.default_rng(0)
for ntaps in (33, 65, 100, 129, 200, 300, 600, 1000):
    for nf in (0, 256, 512, 1024, 2048, 4096):
        if nf and nf < 2 * ntaps: continue
        if nf: os.environ["MI355_FILTER_FFT"] = str(nf)
        else: os.environ.pop("MI355_FILTER_FFT", None)
        taps = rng.standard_normal(ntaps).astype(np.float32)
        blk = pkg.clFilter(1, 2, 0, 0, 1, taps, 1, 0, False)
        nout = n - ntaps
        dt = timeit(lambda: blk.work_device(nout, [x], [y]))
        print("ntaps=%4d NF=%4d (%s): %7.1f GS/s" % (ntaps, blk.fftsize(), "auto" if not nf else "forced", nout / dt / 1e9))
