"""GNU Radio runs every block's work() on its own thread: several blocks of one process work concurrently on the same GPU."""
import threading

import numpy as np
import pytest

from conftest import GPU_ARGS, crandn, relerr

pytestmark = pytest.mark.gpu


def test_blocks_work_concurrently_from_threads(gpu, oracle):
    rng = np.random.default_rng(0)
    errs = []

    def run_fft():
        x = crandn(rng, 8 * 4096)
        y = np.empty_like(x)
        blk = gpu.clFFT(4096, gpu.CLFFT_FORWARD, [], gpu.DTYPE_COMPLEX, *GPU_ARGS)
        ref = oracle.fft_block(4096, True, None, False, oracle.DTYPE_COMPLEX, x, f64=True)
        for _ in range(100):
            blk.work(8, [x], [y])
            if relerr(y, ref) > 1e-5:
                errs.append("fft")

    def run_filter():
        taps = oracle.firdes_low_pass(1.0, 10e6, 1e6, 372000.0)
        x = crandn(rng, 32768 + 64)
        y = np.empty(32768, np.complex64)
        blk = gpu.clFilter(*GPU_ARGS, 1, taps)
        ref = oracle.fir_ccf(taps, x, 32768)
        for _ in range(100):
            blk.work(32768, [x], [y])
            if relerr(y, ref) > 1e-5:
                errs.append("filter")

    def run_math():
        a = crandn(rng, 8192)
        c = np.empty_like(a)
        blk = gpu.clMathOp(gpu.DTYPE_COMPLEX, *GPU_ARGS, gpu.MATHOP_MULTIPLY)
        for _ in range(200):
            blk.work(8192, [a, a], [c])
            if not np.allclose(c, a * a, rtol=1e-6):
                errs.append("math")

    def run_xe():
        N, F, T = 8, 16, 64
        x = np.random.default_rng(5).integers(-128, 128, size=T * N * F * 2, dtype=np.int64).astype(np.int8)
        blk = gpu.clXEngine(*GPU_ARGS, False, gpu.DTYPE_BYTE, 1, N, 1, 0, F, T, [])
        out = np.empty(blk.get_output_buffer_size(), np.complex64)
        ref = oracle.xengine_ichar(N, F, 1, T, x, exact=True)
        for _ in range(100):
            blk.xcorrelate(x, out)
            if not np.array_equal(out, ref):
                errs.append("xengine")

    def run_pfb():
        M, buf = 64, 64 * 64
        taps = np.random.default_rng(6).standard_normal(M * 8).astype(np.float32)
        x = crandn(np.random.default_rng(7), buf + taps.size - M)
        y = np.empty(buf, np.complex64)
        blk = gpu.clPolyphaseChannelizer(*GPU_ARGS, taps, buf, M, M, list(range(M)))
        ref = oracle.pfb(taps, buf, M, M, list(range(M)), x, f64=True)
        for _ in range(100):
            blk.general_work(buf, [x.size], [x], [y])
            if relerr(y, ref) > 1e-5:
                errs.append("pfb")

    big = gpu.clFFT(65536, gpu.CLFFT_FORWARD, [], gpu.DTYPE_COMPLEX, *GPU_ARGS)  # ONE handle, two threads: its workspace is shared
    xb = crandn(np.random.default_rng(8), 3 * 65536)
    refb = oracle.fft_block(65536, True, None, False, oracle.DTYPE_COMPLEX, xb, f64=True)

    def run_big_fft():
        y = np.empty_like(xb)
        for _ in range(30):
            big.work(3, [xb], [y])
            if relerr(y, refb) > 1e-5:
                errs.append("fft65536")

    def run_large_copy():  # 2^22 items: the pinned double buffers and the helper pool of the host path
        a = crandn(np.random.default_rng(9), 1 << 22)
        c = np.empty_like(a)
        blk = gpu.clMathConst(gpu.DTYPE_COMPLEX, *GPU_ARGS, 3.0, gpu.MATHOP_MULTIPLY)
        for _ in range(10):
            blk.work(a.size, [a], [c])
            if not np.array_equal(c, np.float32(3.0) * a):
                errs.append("large copy")

    def run_mixed_radix():  # two threads make a block of the same 2-3-5-7 length at once: its workgroup shape is being measured meanwhile
        n = 1500
        x = crandn(np.random.default_rng(10), 5 * n)
        y = np.empty_like(x)
        blk = gpu.clFFT(n, gpu.CLFFT_FORWARD, [], gpu.DTYPE_COMPLEX, *GPU_ARGS)
        ref = oracle.fft_block(n, True, None, False, oracle.DTYPE_COMPLEX, x, f64=True)
        for _ in range(50):
            blk.work(5, [x], [y])
            if relerr(y, ref) > 1e-5:
                errs.append("fft1500")

    ts = [threading.Thread(target=f) for f in (run_fft, run_filter, run_math, run_xe, run_fft, run_math, run_pfb, run_big_fft, run_big_fft,
                                               run_large_copy, run_large_copy, run_mixed_radix, run_mixed_radix)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs, errs[:5]
