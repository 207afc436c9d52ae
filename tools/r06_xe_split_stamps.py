#!/usr/bin/env python3
"""Per-workgroup phase stamps of the time-range form (MI355_XE_TS=1: synchronous launches, summary on stderr), inputs in rotation.
usage: python tools/r06_xe_split_stamps.py [windows]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import __graft_entry__ as e  # noqa: E402

pkg = e.load_package()
N, F, T = 64, 1024, 1024
nint = int(sys.argv[1]) if len(sys.argv) > 1 else 1
xe = pkg.clXEngine(1, 2, 0, 0, False, pkg.DTYPE_BYTE, 1, N, 1, 0, F, T, [])
g = torch.Generator(device="cuda").manual_seed(1)
bufs = [torch.randint(-127, 128, (nint, T, N, F, 1, 2), dtype=torch.int8, device="cuda", generator=g) for _ in range(6)]
vis = torch.zeros(nint * xe.get_output_buffer_size(), 2, device="cuda")


def fn(i):
    if nint == 1: xe.xcorrelate_device(bufs[i % 6], vis)
    else: xe.xcorrelate_n_device(nint, bufs[i % 6], vis)


for i in range(20): fn(i)
torch.cuda.synchronize()
os.environ["MI355_XE_TS"] = "1"
for i in range(4): fn(i)
