#!/usr/bin/env python3
"""ONE BASELINE.json config per process, at the size bench.py quotes it on -- so that a rocprofv3 trace of this command
averages launches of one size only (bench.py's own trace mixes k_ols / k_xe_i8_lines launches of several sizes).

usage: python tools/baseline_cfg.py <2|3|4|5|5b> [launches]
  2   clFFT 4096 forward, Blackman window + shift, 16384 frames per launch        (k_fft<4096,...>)
  3   clFilter FFT mode, 65 taps, decimation 1, 2^26 - 64 samples per launch      (k_ols<256>)
  4   clPolyphaseChannelizer 64 channels x 32 taps per arm, 2^26 - 2^16 items     (k_pfbw<64,32,...>)
  5   clXEngine 64 antennas x 1024 channels x 1024 frames IChar, one window per call, inputs in rotation (read from HBM)
  5b  the same, 8 windows per launch (mi355_xengine_xcorrelate_n_dev)
Prints one line: config, HIP-event us per launch, algorithmic bytes per launch, fraction of 8 TB/s.
tools/make_profiles_cfg.sh wraps it in the kernel-trace and the two PMC passes; tools/collect_profiles_cfg.py writes profiles/<tag>_baseline_configs.txt.
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import bench  # noqa: E402  (lowpass_taps: the product-independent tap design)
import __graft_entry__ as e  # noqa: E402

pkg = e.load_package()
ARGS = (1, 2, 0, 0)
cfg = sys.argv[1]
launches = int(sys.argv[2]) if len(sys.argv) > 2 else 100
n = 1 << 26


def timed(fn, k):
    # A device that has just been idle runs its first launches 5-10 % slower (the same FFT launch: 199 us after a second of device copies, 181 us
    # behind bench.py's 5 s leg, one box): BASELINE_CFG_WARM_S seconds (default 3) of the config's own launch first, untimed.  In a kernel trace they
    # are in the average too -- at 3 s they outnumber the cold first launches a thousand to one.  (The PMC passes run with 0.2 s: counters serialise.)
    import time
    warm = float(os.environ.get("BASELINE_CFG_WARM_S", "3"))
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < warm:
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(k):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e-3 / k


if cfg == "2":
    nk = np.arange(4096)
    w = (0.42 - 0.5 * np.cos(2 * np.pi * nk / 4095) + 0.08 * np.cos(4 * np.pi * nk / 4095)).astype(np.float32)
    x = torch.randn(16384 * 4096, 2, device="cuda")
    y = torch.empty_like(x)
    blk = pkg.clFFT(4096, pkg.CLFFT_FORWARD, w, pkg.DTYPE_COMPLEX, *ARGS, 0, 1, True)
    dt = timed(lambda: blk.work_device(16384, [x], [y]), launches)
    alg, what = 16384 * 4096 * 16, "clFFT 4096 fwd + blackman + shift, 16384 frames per launch"
elif cfg == "3":
    taps = bench.lowpass_taps(1.0, 10e6, 1e6, 372000.0)
    x = torch.randn(n, 2, device="cuda")
    y = torch.empty_like(x)
    nf = n - 64
    blk = pkg.clFilter(*ARGS, 1, taps, 1, 0, False)
    dt = timed(lambda: blk.work_device(nf, [x], [y]), launches)
    alg, what = nf * 16, "clFilter FFT mode, 65 taps, decim 1, %d samples per launch (fft size %d)" % (nf, blk.fftsize())
elif cfg == "4":
    taps = np.concatenate([bench.lowpass_taps(1.0, 64.0, 0.5, 0.0753), [0.0]]).astype(np.float32)
    buf = (1 << 26) - (1 << 16)
    blk = pkg.clPolyphaseChannelizer(*ARGS, taps, buf, 64, 64, list(range(64)))
    x = torch.randn(n, 2, device="cuda")
    y = torch.empty_like(x)
    xi, yo = x[:blk.ninput()], y[:blk.noutput()]
    dt = timed(lambda: blk.work_device([xi], [yo]), launches)
    alg, what = buf * 16, "clPolyphaseChannelizer 64 x 32, %d items per launch" % buf
elif cfg in ("5", "5b"):
    N, F, T = 64, 1024, 1024
    nint = 8 if cfg == "5b" else 1
    xe = pkg.clXEngine(*ARGS, False, pkg.DTYPE_BYTE, 1, N, 1, 0, F, T, [])
    g = torch.Generator(device="cuda").manual_seed(42)
    per = T * N * F * 2
    k = max(2, int(-(-640e6 // (nint * per))) + 1)
    bufs = [torch.randint(-127, 128, (nint, T, N, F, 1, 2), dtype=torch.int8, device="cuda", generator=g) for _ in range(k)]
    vis = torch.zeros(nint * xe.get_output_buffer_size(), 2, device="cuda")
    st = [0]

    def fn():
        xb = bufs[st[0] % k]
        st[0] += 1
        if nint == 1:
            xe.xcorrelate_device(xb, vis)
        else:
            xe.xcorrelate_n_device(nint, xb, vis)
    dt = timed(fn, launches)
    alg = nint * (per + xe.get_output_buffer_size() * 8)
    what = "clXEngine 64 ant x 1024 ch x 1024 frames IChar, %d window(s) per launch, %d inputs in rotation" % (nint, k)
else:
    raise SystemExit(__doc__)
print(json.dumps({"config": cfg, "what": what, "launches": launches, "us_per_launch": round(dt * 1e6, 2), "algorithmic_bytes_per_launch": alg,
                  "hbm_frac": round(alg / dt / 8e12, 4)}), flush=True)
