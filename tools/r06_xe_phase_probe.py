import os, sys
sys.path.insert(0, "/root/repo")
import torch, __graft_entry__ as e
pkg = e.load_package()
N, F, T = 64, 1024, 1024
xe = pkg.clXEngine(1, 2, 0, 0, False, pkg.DTYPE_BYTE, 1, N, 1, 0, F, T, [])
g = torch.Generator(device="cuda").manual_seed(1)
per = xe.get_output_buffer_size()
for nint in (1, 8):
    k = max(2, int(-(-640e6 // (nint * T * N * F * 2))) + 1)
    bufs = [torch.randint(-127, 128, (nint, T, N, F, 1, 2), dtype=torch.int8, device="cuda", generator=g) for _ in range(k)]
    vis = torch.zeros(nint * per, 2, device="cuda")
    for name, dbg in (("default", "0"), ("no DMA", "4"), ("no matrix stores", "2"), ("no DMA, no stores", "6")):
        os.environ["MI355_XE_DBG"] = dbg
        def fn(i):
            if nint == 1: xe.xcorrelate_device(bufs[i % k], vis)
            else: xe.xcorrelate_n_device(nint, bufs[i % k], vis)
        for i in range(10): fn(i)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for i in range(200): fn(i)
        b.record(); torch.cuda.synchronize()
        print("%d window(s): %-20s %.1f us per window" % (nint, name, a.elapsed_time(b) * 1e3 / 200 / nint), flush=True)
    del bufs, vis; torch.cuda.empty_cache()
