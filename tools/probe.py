#!/usr/bin/env python3
"""GPU probes (tuning aids; not part of the product or of the test suite).  One script, sub-commands:

  rates   [block ...]            device-resident rate of a block over its usual parameter range
                                 blocks: math fft fftreal filter fir longfilter pfb xengine xcorr elem
  ab      <case> '<json list>'   interleaved A/B of environment-variable variants inside ONE process (the drift between
                                 processes is +-4 %); cases: fft<N> filter65 fir65 filter3000 pfb mathconst xengine
  latency                        host-pointer work() calls at scheduler sizes (us per call)
  host                           host-pointer work() throughput of large calls (PCIe inclusive)
  pfbsmall                       the channelizer at the reference's call sizes: eager, HIP-graph replay, batched
Environment: PROBE_LOG2 (log2 of the samples per call, default 26 = 512 MiB per buffer, past the 256 MiB Infinity Cache).
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import __graft_entry__ as e  # noqa: E402

pkg = e.load_package()
o = e.load_oracle()
ARGS = (1, 2, 0, 0)
LOG2 = int(os.environ.get("PROBE_LOG2", "26"))
N = 1 << LOG2


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


def show(label, dt, samples, bytes_per_sample=16, extra=""):
    print("%-58s %8.1f us %8.1f GS/s %6.2f TB/s (%4.1f %% of 8 TB/s) %s" % (label, dt * 1e6, samples / dt / 1e9, samples * bytes_per_sample / dt / 1e12,
                                                                             samples * bytes_per_sample / dt / 8e10, extra), flush=True)


def bufs():
    return torch.randn(N, 2, device="cuda"), torch.empty(N, 2, device="cuda")


def taps65():
    return o.firdes_low_pass(1.0, 10e6, 1e6, 372000.0)


def taps2048():
    return np.concatenate([o.firdes_low_pass(1.0, 64.0, 0.5, 0.0753), [0.0]]).astype(np.float32)


# ---- rates -----------------------------------------------------------------------------------------------------------
def rates_math():
    a, c = bufs()
    b = torch.randn_like(a)
    for op in ("MULTIPLY", "ADD", "MULTIPLY_CONJUGATE"):
        blk = pkg.clMathOp(pkg.DTYPE_COMPLEX, *ARGS, getattr(pkg, "MATHOP_" + op))
        show("clMathOp complex " + op, ev_time(lambda: blk.work_device(N, [a, b], [c])), N, 24)
    blk = pkg.clMathConst(pkg.DTYPE_COMPLEX, *ARGS, 2.0, pkg.MATHOP_MULTIPLY)
    show("clMathConst complex MULTIPLY", ev_time(lambda: blk.work_device(N, [a], [c])), N, 16)


def rates_fft(real=False):
    a, c = bufs()
    for lg in list(range(1, 17)) + ["1000", "1536", "6000"]:
        n = int(lg) if isinstance(lg, str) else 1 << lg
        w = np.blackman(n).astype(np.float32)
        blk = pkg.clFFT(n, pkg.CLFFT_FORWARD, w, pkg.DTYPE_FLOAT if real else pkg.DTYPE_COMPLEX, *ARGS, 0, 1, True)
        nv = (N // 2 if real else N) // n
        x = a.view(-1)[:nv * n] if real else a
        show("clFFT %s N=%d window+shift" % ("real" if real else "complex", n), ev_time(lambda: blk.work_device(nv, [x], [c])), nv * n, 12 if real else 16)


def rates_filter(use_time):
    a, c = bufs()
    rng = np.random.default_rng(1)
    for nt in (3, 16, 65, 129, 300, 497, 1000, 2000):
        t = (rng.standard_normal(nt) / np.sqrt(nt)).astype(np.float32)
        for dec in (1, 4):
            blk = pkg.clFilter(*ARGS, dec, t, 1, 0, use_time)
            nout = (N - nt) // dec
            dt = ev_time(lambda: blk.work_device(nout, [a], [c]), iters=5 if use_time and nt > 300 else 20)
            show("clFilter %s %4d taps decim %d (fft %s)" % ("direct" if use_time else "fast-conv", nt, dec, blk.fftsize() if not use_time else "-"), dt, nout * dec, 16)


def rates_longfilter():
    a, c = bufs()
    rng = np.random.default_rng(2)
    for nt in (2049, 3000, 4096, 6000, 8192, 10240, 16384):
        t = (rng.standard_normal(nt) / np.sqrt(nt)).astype(np.float32)
        blk = pkg.clFilter(*ARGS, 1, t, 1, 0, False)
        show("clFilter fast-conv %5d taps (partitioned)" % nt, ev_time(lambda: blk.work_device(N - nt, [a], [c]), iters=100, warm=20), N - nt, 16)


def rates_pfb():
    a, c = bufs()
    t = taps2048()
    for M in (2, 4, 8, 16, 32, 64, 128, 256):
        for tpa in (8, 32):
            tp = np.resize(t, M * tpa).astype(np.float32)
            buf = (N - (1 << 16)) // M * M
            blk = pkg.clPolyphaseChannelizer(*ARGS, tp, buf, M, M, list(range(M)))
            show("clPolyphaseChannelizer M=%3d taps/arm=%2d" % (M, tpa), ev_time(lambda: blk.work_device([a], [c])), buf, 16)


def rates_xengine():
    for (Na, F, T, npol, dt_, name) in ((64, 1024, 1024, 1, pkg.DTYPE_BYTE, "ichar"), (32, 1024, 1024, 2, pkg.DTYPE_BYTE, "ichar dual-pol"),
                                        (64, 1024, 1024, 2, pkg.DTYPE_PACKEDXY, "packed 4-bit"), (64, 1024, 1024, 1, pkg.DTYPE_COMPLEX, "cf32")):
        if dt_ == pkg.DTYPE_COMPLEX:
            x = torch.randn(T * Na * F * npol, 2, device="cuda")
        elif dt_ == pkg.DTYPE_PACKEDXY:
            x = torch.randint(-128, 128, (T * Na * F * 2,), dtype=torch.int8, device="cuda")
        else:
            x = torch.randint(-127, 128, (T * Na * F * npol * 2,), dtype=torch.int8, device="cuda")
        blk = pkg.clXEngine(*ARGS, False, dt_, npol, Na, 1, 0, F, T, [])
        out = torch.zeros(blk.get_output_buffer_size(), 2, device="cuda")
        # inputs in rotation (>= 640 MB in all): a 134 MB input alone would sit in the 256 MiB Infinity Cache between launches
        xs = [x] + [x.clone() for _ in range(max(1, int(640e6 // (x.numel() * x.element_size()))))]
        turn = [0]

        def call():
            blk.xcorrelate_device(xs[turn[0] % len(xs)], out)
            turn[0] += 1
        dt = ev_time(call)
        nb = Na * (Na + 1) // 2
        alg = x.numel() * x.element_size() + out.numel() * 4
        flop = 8.0 * F * nb * T * (2 if dt_ == pkg.DTYPE_PACKEDXY else npol) ** 2
        print("clXEngine %-16s N=%d F=%d T=%d npol=%d: %7.1f us  %6.1f Top/s  algorithmic %.2f TB/s (%.1f %% of 8 TB/s)" % (
            name, Na, F, T, npol, dt * 1e6, flop / dt / 1e12, alg / dt / 1e12, alg / dt / 8e10), flush=True)


def rates_xcorr():
    a, c = bufs()
    for n, nin in ((256, 4), (1024, 4), (4096, 2), (1024, 16)):
        fr = (N // nin) // n
        blk = pkg.clxcorrelate_fft_vcf(n, nin, *ARGS, 2)
        xi = [a[i * fr * n:(i + 1) * fr * n] for i in range(nin)]
        xo = [c.view(-1)[i * fr * n:(i + 1) * fr * n] for i in range(nin - 1)]
        show("clxcorrelate_fft_vcf N=%d inputs=%d" % (n, nin), ev_time(lambda: blk.work_device(fr, xi, xo)), nin * fr * n, (8 * nin + 4 * (nin - 1)) / nin)


def rates_elem():
    a, c = bufs()
    af = a.view(-1)[:N].abs() + 0.1
    cf = c.view(-1)
    for name, ctor, ins, outs, bps in (("clLog", lambda: pkg.clLog(*ARGS, 10.0, 0.0), [af], [cf], 8), ("clComplexToMag", lambda: pkg.clComplexToMag(*ARGS), [a], [cf], 12),
                                       ("clComplexToArg", lambda: pkg.clComplexToArg(*ARGS), [a], [cf], 12),
                                       ("clQuadratureDemod", lambda: pkg.clQuadratureDemod(1.0, *ARGS), [a], [cf], 12),
                                       ("clComplexToMagPhase", lambda: pkg.clComplexToMagPhase(*ARGS), [a], [cf[:N], cf[N:2 * N]], 16),
                                       ("clMagPhaseToComplex", lambda: pkg.clMagPhaseToComplex(*ARGS), [af, a.view(-1)[N:2 * N]], [c], 16),
                                       ("clSNR", lambda: pkg.clSNR(*ARGS, 10.0, 0.0), [af, af], [cf], 12)):
        blk = ctor()
        n = N - 1
        show(name, ev_time(lambda: blk.work_device(n, ins, outs)), n, bps)


RATES = {"math": rates_math, "fft": rates_fft, "fftreal": lambda: rates_fft(True), "filter": lambda: rates_filter(False), "fir": lambda: rates_filter(True),
         "longfilter": rates_longfilter, "pfb": rates_pfb, "xengine": rates_xengine, "xcorr": rates_xcorr, "elem": rates_elem}


# ---- ab --------------------------------------------------------------------------------------------------------------
def ab(case, variants):
    a, c = bufs()
    if case == "filter65":
        blk = pkg.clFilter(*ARGS, 1, taps65(), 1, 0, False); f = lambda: blk.work_device(N - 64, [a], [c])
    elif case == "fir65":
        blk = pkg.clFilter(*ARGS, 1, taps65(), 1, 0, True); f = lambda: blk.work_device(N - 64, [a], [c])
    elif case == "filter3000":
        t = (np.random.default_rng(3000).standard_normal(3000) / 55).astype(np.float32)
        blk = pkg.clFilter(*ARGS, 1, t, 1, 0, False); f = lambda: blk.work_device(N - 3000, [a], [c])
    elif case == "pfb":
        buf = N - (1 << 16)
        blk = pkg.clPolyphaseChannelizer(*ARGS, taps2048(), buf, 64, 64, list(range(64))); f = lambda: blk.work_device([a], [c])
    elif case == "mathconst":
        blk = pkg.clMathConst(pkg.DTYPE_COMPLEX, *ARGS, 2.0, pkg.MATHOP_MULTIPLY); f = lambda: blk.work_device(N, [a], [c])
    elif case == "xengine":
        x = torch.randint(-127, 128, (1024 * 64 * 1024 * 2,), dtype=torch.int8, device="cuda")
        out = torch.zeros(1024 * 2080, 2, device="cuda")
        blks = {}

        def f():  # the plan is fixed at construction: one block per variant
            key = json.dumps({k: os.environ.get(k) for k in keys})
            if key not in blks:
                blks[key] = pkg.clXEngine(*ARGS, False, pkg.DTYPE_BYTE, 1, 64, 1, 0, 1024, 1024, [])
            blks[key].xcorrelate_device(x, out)
    elif case == "xengine_cf32":
        x = torch.randn(1024 * 64 * 1024, 2, device="cuda")
        blk = pkg.clXEngine(*ARGS, False, pkg.DTYPE_COMPLEX, 1, 64, 1, 0, 1024, 1024, [])
        out = torch.zeros(blk.get_output_buffer_size(), 2, device="cuda")
        f = lambda: blk.xcorrelate_device(x, out)
    elif case.startswith("fft"):
        n = int(case[3:])
        blk = pkg.clFFT(n, pkg.CLFFT_FORWARD, np.blackman(n).astype(np.float32), pkg.DTYPE_COMPLEX, *ARGS, 0, 1, True); f = lambda: blk.work_device(N // n, [a], [c])
    else:
        raise SystemExit("unknown case " + case)
    keys = sorted({k for v in variants for k in v})
    res = [[] for _ in variants]
    for _ in range(100):
        f()
    for _ in range(8):
        for i, v in enumerate(variants):
            for k in keys:
                os.environ.pop(k, None)
            os.environ.update(v)
            res[i].append(ev_time(f, iters=50, warm=10) * 1e6)
    for v, r in zip(variants, res):
        r2 = sorted(r)
        print("%-12s %-64s median %8.2f us  min %8.2f" % (case, json.dumps(v), r2[len(r2) // 2], r2[0]), flush=True)


# ---- host-pointer paths ----------------------------------------------------------------------------------------------
def latency():
    rng = np.random.default_rng(0)

    def lat(fn, iters=300):
        for _ in range(20):
            fn()
        t0 = time.perf_counter()
        for _ in range(iters):
            fn()
        return (time.perf_counter() - t0) / iters * 1e6

    def crandn(n):
        return (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)

    for n in (1024, 8192, 32768, 65536, 262144):
        a, b, c = crandn(n), crandn(n), np.empty(n, np.complex64)
        mul = pkg.clMathOp(pkg.DTYPE_COMPLEX, *ARGS, pkg.MATHOP_MULTIPLY)
        t = lat(lambda: mul.work(n, [a, b], [c]))
        print("clMathOp.work %7d items: %7.1f us/call %8.1f MS/s" % (n, t, n / t))
    for n in (8192, 32768, 65536):
        x, y = crandn(n + 64), np.empty(n, np.complex64)
        f = pkg.clFilter(*ARGS, 1, taps65(), 1, 0, False)
        t = lat(lambda: f.work(n, [x], [y]))
        print("clFilter(fft,65).work %6d items: %7.1f us/call %8.1f MS/s" % (n, t, n / t))
    for nvec in (1, 2, 8):
        x = crandn(nvec * 4096); y = np.empty_like(x)
        f = pkg.clFFT(4096, pkg.CLFFT_FORWARD, np.blackman(4096).astype(np.float32), pkg.DTYPE_COMPLEX, *ARGS, 0, 1, True)
        t = lat(lambda: f.work(nvec, [x], [y]))
        print("clFFT(4096).work %d vectors: %7.1f us/call %8.1f MS/s" % (nvec, t, nvec * 4096 / t))
    for buf in (8192, 65536):
        p = pkg.clPolyphaseChannelizer(*ARGS, taps2048(), buf, 64, 64, list(range(64)))
        x, y = crandn(p.ninput()), np.empty(p.noutput(), np.complex64)
        t = lat(lambda: p.general_work(buf, None, [x], [y]))
        print("clPolyphaseChannelizer.work buf_items=%6d: %7.1f us/call %8.1f MS/s" % (buf, t, buf / t))


def host():
    rng = np.random.default_rng(0)
    for logn in (20, 24, 26):
        n = 1 << logn
        x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64); y = np.empty_like(x)
        fft = pkg.clFFT(4096, pkg.CLFFT_FORWARD, np.blackman(4096).astype(np.float32), pkg.DTYPE_COMPLEX, *ARGS, 0, 1, True)
        mul = pkg.clMathConst(1, *ARGS, 2.0, pkg.MATHOP_MULTIPLY)
        flt = pkg.clFilter(*ARGS, 1, rng.standard_normal(65).astype(np.float32))
        xf = np.concatenate([np.zeros(64, np.complex64), x])

        def rate(fn):
            fn(); t0 = time.perf_counter()
            for _ in range(5):
                fn()
            return n * 5 / (time.perf_counter() - t0) / 1e9
        print("n=2^%d: clFFT %.2f GS/s  clMathConst %.2f GS/s  clFilter %.2f GS/s (8 B in + 8 B out per sample over PCIe)" % (
            logn, rate(lambda: fft.work(n // 4096, [x], [y])), rate(lambda: mul.work(n, [x], [y])), rate(lambda: flt.work(n, [xf], [y]))))


def pfbsmall():
    t = taps2048()
    for buf in (8192, 65536, 262144):
        p = pkg.clPolyphaseChannelizer(*ARGS, t, buf, 64, 64, list(range(64)))
        x = torch.randn(8 * buf + 2048 - 64, 2, device="cuda"); y = torch.empty(8 * buf, 2, device="cuda")
        eager = ev_time(lambda: p.work_device([x], [y]), iters=400, warm=20)
        g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            with torch.cuda.graph(g, stream=s):
                for _ in range(256):
                    p.work_device([x], [y])
        torch.cuda.synchronize()
        graph = ev_time(g.replay, iters=10, warm=2) / 256
        batched = ev_time(lambda: p.work_device([x], [y], nbuf=8), iters=200, warm=10) / 8
        print("buf_items=%6d: eager %.2f us/launch (host launch rate), graph replay %.2f us/launch, batched x8 %.2f us/buffer" % (
            buf, eager * 1e6, graph * 1e6, batched * 1e6))


if __name__ == "__main__":
    cmd = sys.argv[1] if len(sys.argv) > 1 else "rates"
    if cmd == "rates":
        for b in (sys.argv[2:] or list(RATES)):
            RATES[b]()
    elif cmd == "ab":
        ab(sys.argv[2], json.loads(sys.argv[3]))
    elif cmd in ("latency", "host", "pfbsmall"):
        {"latency": latency, "host": host, "pfbsmall": pfbsmall}[cmd]()
    else:
        raise SystemExit(__doc__)
