import os, sys
sys.path.insert(0, "/root/repo")
import torch, __graft_entry__ as e
pkg = e.load_package()
N, F, T = 64, 1024, 1024
xe = pkg.clXEngine(1, 2, 0, 0, False, pkg.DTYPE_BYTE, 1, N, 1, 0, F, T, [])
g = torch.Generator(device="cuda").manual_seed(1)
per = xe.get_output_buffer_size()
for nint in (4, 8, 32):
    k = max(2, int(-(-640e6 // (nint * T * N * F * 2))) + 1)
    bufs = [torch.randint(-127, 128, (nint, T, N, F, 1, 2), dtype=torch.int8, device="cuda", generator=g) for _ in range(k)]
    vis = torch.zeros(nint * per, 2, device="cuda")
    launches = max(20, int(0.08 / (nint * 37e-6)))
    for pf in ("0", "2", "3", "4", "6", "8"):
        os.environ["MI355_XE_LINES_PF"] = pf
        ts = []
        for _ in range(2):
            for i in range(10): xe.xcorrelate_n_device(nint, bufs[i % k], vis)
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for i in range(launches): xe.xcorrelate_n_device(nint, bufs[i % k], vis)
            b.record(); torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) * 1e3 / launches / nint)
        print("%2d windows, pace 2, touches %s ahead: %s" % (nint, pf, " ".join("%.2f" % t for t in ts)), flush=True)
    del bufs, vis; torch.cuda.empty_cache()
