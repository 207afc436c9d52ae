#!/usr/bin/env python3
"""Config 5, one / two / three windows per call, inputs in rotation (read from HBM): the time-range form of the whole-line kernel against the
32-byte-slice kernel, interleaved in ONE process (environment switches are read per launch).  usage: python tools/r06_xe_split_probe.py [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import __graft_entry__ as e  # noqa: E402

pkg = e.load_package()
N, F, T = 64, 1024, 1024
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
xe = pkg.clXEngine(1, 2, 0, 0, False, pkg.DTYPE_BYTE, 1, N, 1, 0, F, T, [])
g = torch.Generator(device="cuda").manual_seed(1)
per = xe.get_output_buffer_size()


def timed(nint, env, launches=300):
    k = max(2, int(-(-640e6 // (nint * T * N * F * 2))) + 1)
    bufs = [torch.randint(-127, 128, (nint, T, N, F, 1, 2), dtype=torch.int8, device="cuda", generator=g) for _ in range(k)]
    vis = torch.zeros(nint * per, 2, device="cuda")
    old = {kk: os.environ.get(kk) for kk in env}
    os.environ.update(env)
    try:
        def fn(i):
            if nint == 1: xe.xcorrelate_device(bufs[i % k], vis)
            else: xe.xcorrelate_n_device(nint, bufs[i % k], vis)
        for i in range(10): fn(i)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for i in range(launches): fn(i)
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) * 1e3 / launches / nint
    finally:
        for kk, v in old.items():
            if v is None: os.environ.pop(kk, None)
            else: os.environ[kk] = v


variants = [("time ranges (default)", {}), ("slice kernel", {"MI355_XE_NO_LINES_SPLIT": "1"}), ("ranges, no touches", {"MI355_XE_LINES_PF": "0"}),
            ("ranges, touches 2 ahead", {"MI355_XE_LINES_PF": "2"}), ("ranges, touches 3 ahead", {"MI355_XE_LINES_PF": "3"}),
            ("ranges, no matrix stores", {"MI355_XE_DBG": "2"})]
if len(sys.argv) > 2:
    variants = variants[:2]
for nint in (1, 2, 3):
    for name, env in variants:
        ts = [timed(nint, env) for _ in range(reps)]
        print("%d window(s) per call  %-28s %s us per window" % (nint, name, " ".join("%.1f" % t for t in ts)), flush=True)
