#!/usr/bin/env python3
"""64 antennas x 2 polarisations x 1024 channels x 1024 frames (the reference CLI's default), inputs in rotation: the whole-line kernel against corner
turn + correlator, interleaved in one process.  usage: python tools/r06_xe_dualpol_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import __graft_entry__ as e  # noqa: E402

pkg = e.load_package()
N, F, T = 64, 1024, 1024
xe = pkg.clXEngine(1, 2, 0, 0, False, pkg.DTYPE_BYTE, 2, N, 1, 0, F, T, [])
g = torch.Generator(device="cuda").manual_seed(1)
bufs = [torch.randint(-127, 128, (T, N, F, 2, 2), dtype=torch.int8, device="cuda", generator=g) for _ in range(4)]
vis = torch.zeros(xe.get_output_buffer_size(), 2, device="cuda")
for name, env in (("whole-line kernel", {}), ("corner turn + correlator", {"MI355_XE_NO_LINES2": "1"}), ("whole-line kernel, no matrix stores", {"MI355_XE_DBG": "2"}),
                  ("whole-line kernel, no DMA", {"MI355_XE_DBG": "4"})):
    os.environ.update(env)
    ts = []
    for _ in range(3):
        for i in range(10): xe.xcorrelate_device(bufs[i % 4], vis)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for i in range(300): xe.xcorrelate_device(bufs[i % 4], vis)
        b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3 / 300)
    for k in env: os.environ.pop(k)
    print("%-40s %s us per integration (%s)" % (name, " ".join("%.1f" % t for t in ts), xe.last_route()["kernel"]), flush=True)
os.environ["MI355_XE_TS"] = "1"
xe.xcorrelate_device(bufs[0], vis)
os.environ.pop("MI355_XE_TS", None)
per = xe.get_output_buffer_size()
for nint in (2, 4, 8):
    xb = [torch.randint(-127, 128, (nint, T, N, F, 2, 2), dtype=torch.int8, device="cuda", generator=g) for _ in range(2)]
    vb = torch.zeros(nint * per, 2, device="cuda")
    for i in range(5): xe.xcorrelate_n_device(nint, xb[i % 2], vb)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(100): xe.xcorrelate_n_device(nint, xb[i % 2], vb)
    b.record(); torch.cuda.synchronize()
    print("%d windows per launch: %.1f us per window (%s)" % (nint, a.elapsed_time(b) * 1e3 / 100 / nint, xe.last_route()), flush=True)
    del xb, vb
