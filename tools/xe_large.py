#!/usr/bin/env python3
"""X-engine rates for the large geometries (rows > 64) and the per-rank / batched forms; run under tools/prof_kernels.sh for the
per-kernel split.  usage: xe_large.py [N,F,T,npol ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import __graft_entry__ as e  # noqa: E402

pkg = e.load_package()
ARGS = (1, 2, 0, 0)


def ev_time(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e-3 / iters


def rotating(x, call):
    """call(x_k) over enough copies of x (>= 640 MB in all) that no launch finds its input in the 256 MiB Infinity Cache"""
    xs = [x] + [x.clone() for _ in range(max(1, int(640e6 // x.numel())))]
    turn = [0]

    def fn():
        call(xs[turn[0] % len(xs)])
        turn[0] += 1
    return fn


cases = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]] or [(96, 1024, 1024, 1), (128, 1024, 1024, 1), (64, 1024, 1024, 2), (256, 512, 1024, 1), (128, 512, 1024, 2)]
for (Na, F, T, npol) in cases:
    x = torch.randint(-127, 128, (T * Na * F * npol * 2,), dtype=torch.int8, device="cuda")
    blk = pkg.clXEngine(*ARGS, False, pkg.DTYPE_BYTE, npol, Na, 1, 0, F, T, [])
    out = torch.zeros(blk.get_output_buffer_size(), 2, device="cuda")
    dt = ev_time(rotating(x, lambda v: blk.xcorrelate_device(v, out)))
    nb = Na * (Na + 1) // 2
    alg = x.numel() + out.numel() * 4
    ops = 8.0 * F * nb * T * npol ** 2
    print("clXEngine ichar N=%d F=%d T=%d npol=%d: %7.1f us  %6.1f Top/s (%.3f of 5 POPS)  algorithmic %.2f TB/s (%.1f %% of 8 TB/s)" % (
        Na, F, T, npol, dt * 1e6, ops / dt / 1e12, ops / dt / 5e15, alg / dt / 1e12, alg / dt / 8e10), flush=True)

# per-rank problem of the 8-GPU sharded config 5 (64 antennas x 128 channels x 1024 frames), one window and batched
if not sys.argv[1:] or os.environ.get("XE_BATCH"):
    Na, F, T = 64, 128, 1024
    blk = pkg.clXEngine(*ARGS, False, pkg.DTYPE_BYTE, 1, Na, 1, 0, F, T, [])
    per = blk.get_output_buffer_size()
    for nint in (1, 2, 4, 8, 16, 32):
        x = torch.randint(-127, 128, (nint * T * Na * F * 2,), dtype=torch.int8, device="cuda")
        out = torch.zeros(nint * per, 2, device="cuda")
        dt = ev_time(rotating(x, lambda v: blk.xcorrelate_n_device(nint, v, out)))
        print("per-rank 64 ant x 128 ch x 1024 t, %2d windows per launch: %7.1f us per launch, %6.2f us per window" % (nint, dt * 1e6, dt * 1e6 / nint), flush=True)
    blk1 = pkg.clXEngine(*ARGS, False, pkg.DTYPE_BYTE, 1, 64, 1, 0, 1024, 1024, [])
    x = torch.randint(-127, 128, (1024 * 64 * 1024 * 2,), dtype=torch.int8, device="cuda")
    out = torch.zeros(blk1.get_output_buffer_size(), 2, device="cuda")
    dt = ev_time(rotating(x, lambda v: blk1.xcorrelate_device(v, out)))
    dt1 = ev_time(lambda: blk1.xcorrelate_device(x, out))
    print("single GPU 64 ant x 1024 ch x 1024 t: %7.1f us (inputs in rotation; one buffer over and over: %.1f us)" % (dt * 1e6, dt1 * 1e6))
