"""Stress run (GPU box; not part of the test suite): blocks are created, used and destroyed from several threads at once, taps are
replaced between device-path calls without an intervening synchronisation, the X-engine runs double-buffered -- every result is
checked against the oracle.  `python tools/stress/stress_blocks.py [seconds]`; prints a summary line, exit code 1 on any mismatch."""
import os, sys, threading, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import torch
import __graft_entry__ as entry

pkg = entry.load_package()
orc = entry.load_oracle()
ARGS = (1, 2, 0, 0)
T_END = time.time() + (float(sys.argv[1]) if len(sys.argv) > 1 else 60.0)
errs, counts, lock = [], {}, threading.Lock()


def crandn(rng, n):
    return (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)


def relerr(a, b):
    return float(np.abs(a - b).max() / max(1e-30, np.abs(b).max()))


def note(name, ok):
    with lock:
        counts[name] = counts.get(name, 0) + 1
        if not ok:
            errs.append(name)


def churn_fft(seed):
    rng = np.random.default_rng(seed)
    while time.time() < T_END:
        n = int(rng.choice([64, 1000, 1500, 2310, 4096, 4099, 8192, 20000, 32768, 65536, 131072]))  # (mixed radix incl. its measurement at create, chirp-z, two-pass forms)
        x = crandn(rng, 2 * n)
        y = np.empty_like(x)
        blk = pkg.clFFT(n, pkg.CLFFT_FORWARD, [], pkg.DTYPE_COMPLEX, *ARGS)
        blk.work(2, [x], [y])
        note("fft create/work/destroy", relerr(y, orc.fft_block(n, True, None, False, orc.DTYPE_COMPLEX, x, f64=True)) < 2e-5)
        del blk


def churn_filter(seed):
    rng = np.random.default_rng(seed)
    while time.time() < T_END:
        nt = int(rng.choice([5, 65, 300, 2500, 5000]))
        taps = (rng.standard_normal(nt) / np.sqrt(nt)).astype(np.float32)
        n = 20000
        x = crandn(rng, n + nt - 1)
        blk = pkg.clFilter(*ARGS, 1, taps, 1, 0, bool(rng.integers(2)) and nt < 600)
        xd = torch.from_numpy(x.view(np.float32).reshape(-1, 2)).cuda()
        yd = torch.empty(n, 2, device="cuda")
        blk.work_device(n, [xd], [yd])
        # new taps right behind an asynchronous device-path call, then another call: both results must be right
        nt2 = int(rng.choice([9, 129, 3000]))
        taps2 = (rng.standard_normal(nt2) / np.sqrt(nt2)).astype(np.float32)
        x2 = crandn(rng, n + nt2 - 1)
        y1 = yd.cpu().numpy().view(np.complex64).reshape(-1).copy() if rng.integers(2) else None
        blk.set_taps2(taps2)
        x2d = torch.from_numpy(x2.view(np.float32).reshape(-1, 2)).cuda()
        y2d = torch.empty(n, 2, device="cuda")
        blk.work_device(n, [x2d], [y2d])
        if y1 is None:
            y1 = yd.cpu().numpy().view(np.complex64).reshape(-1)
        y2 = y2d.cpu().numpy().view(np.complex64).reshape(-1)
        note("filter set_taps between device calls", relerr(y1, orc.fir_ccf(taps, x, n)) < 2e-5 and relerr(y2, orc.fir_ccf(taps2, x2, n)) < 2e-5)
        del blk


def churn_xengine(seed):
    rng = np.random.default_rng(seed)
    while time.time() < T_END:
        N, F, T = [(8, 16, 64), (16, 64, 128), (20, 128, 96), (64, 64, 256)][int(rng.integers(4))]
        blk = pkg.clXEngine(*ARGS, False, pkg.DTYPE_BYTE, 1, N, 1, 0, F, T, [])
        xs = [rng.integers(-128, 128, size=T * N * F * 2, dtype=np.int64).astype(np.int8) for _ in range(4)]
        refs = [orc.xengine_ichar(N, F, 1, T, x, exact=True) for x in xs]
        out = np.empty(blk.get_output_buffer_size(), np.complex64)
        ok = True
        blk.submit(xs[0])
        for i in range(1, 4):  # two integrations in flight
            blk.submit(xs[i])
            blk.wait(out)
            ok = ok and np.array_equal(out, refs[i - 1])
        blk.wait(out)
        ok = ok and np.array_equal(out, refs[3])
        note("xengine double-buffered submit/wait", ok)
        del blk


def churn_pfb_math(seed):
    rng = np.random.default_rng(seed)
    while time.time() < T_END:
        M = int(rng.choice([4, 32, 64, 128, 20, 100, 10, 48, 360, 512]))  # (20 ... 360: filters + mixed-radix transform in one kernel; 512: two kernels)
        tpa = int(rng.choice([3, 8, 32]))
        buf = M * 128
        taps = rng.standard_normal(M * tpa).astype(np.float32)
        x = crandn(rng, buf + taps.size - M)
        y = np.empty(buf, np.complex64)
        blk = pkg.clPolyphaseChannelizer(*ARGS, taps, buf, M, M, list(range(M)))
        blk.general_work(buf, [x.size], [x], [y])
        note("pfb create/work/destroy", relerr(y, orc.pfb(taps, buf, M, M, list(range(M)), x, f64=True)) < 2e-5)
        if M >= 64 and M <= 128:  # 2- / 4-fold oversampled on the ring kernel (one launch per residue of the step number)
            R = M // int(rng.choice([2, 4]))
            xo = crandn(rng, buf - R + taps.size)
            blk2 = pkg.clPolyphaseChannelizer(*ARGS, taps, buf, M, R, list(range(M)))
            yo = np.empty(blk2.noutput(), np.complex64)
            blk2.general_work(yo.size, [xo.size], [xo], [yo])
            note("pfb oversampled create/work/destroy", relerr(yo, orc.pfb(taps, buf, M, R, list(range(M)), xo, f64=True)) < 2e-5)
            del blk2
        a = crandn(rng, 1 << int(rng.integers(10, 21)))
        c = np.empty_like(a)
        mc = pkg.clMathConst(pkg.DTYPE_COMPLEX, *ARGS, 1.5, pkg.MATHOP_MULTIPLY)
        mc.work(a.size, [a], [c])
        note("mathconst host path", np.array_equal(c, np.float32(1.5) * a))
        del blk, mc


ths = [threading.Thread(target=f, args=(i,)) for i, f in enumerate((churn_fft, churn_fft, churn_filter, churn_filter, churn_xengine, churn_xengine, churn_pfb_math))]
[t.start() for t in ths]
[t.join() for t in ths]
print("stress:", counts, "mismatches:", errs[:10], flush=True)
sys.exit(1 if errs else 0)
