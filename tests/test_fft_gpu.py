"""GPU parity: clFFT through the C ABI vs the oracle and the golden vectors.

Tolerance (DESIGN.md 'Tolerances'): max|got-ref| <= 1e-5 * max|ref| per call, ref =
float64 DFT rounded to float; single-precision FFTs sit near 3e-7 on this metric."""
import numpy as np
import pytest

from conftest import GPU_ARGS, crandn, golden, relerr

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _fft(gpu, n, direction, window=None, dtype=None, nstreams=1, shift=False):
    dtype = gpu.DTYPE_COMPLEX if dtype is None else dtype
    return gpu.clFFT(n, direction, [] if window is None else window, dtype, *GPU_ARGS, 0, nstreams, shift)


@pytest.mark.parametrize("n", [2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384])
@pytest.mark.parametrize("fwd", [True, False])
def test_all_sizes_both_directions(gpu, oracle, n, fwd):
    rng = np.random.default_rng(n)
    nvec = max(3, 3 * 4096 // n + 1)  # ragged: not a multiple of the frames-per-workgroup
    x = crandn(rng, nvec * n)
    y = np.empty_like(x)
    blk = _fft(gpu, n, gpu.CLFFT_FORWARD if fwd else gpu.CLFFT_BACKWARD)
    assert blk.work(nvec, [x], [y]) == nvec
    ref = oracle.fft_block(n, fwd, None, False, oracle.DTYPE_COMPLEX, x, f64=True)
    assert relerr(y, ref) <= TOL
    # and it is as accurate as a float FFT should be, not merely inside the budget
    assert relerr(y, ref) <= 2e-6


def test_reference_tone_known_answer(gpu):
    # lib/clFFT_impl.cc:361-455: N=2048 one-cycle tone -> X[2047] = (0, 2048)
    x = golden("fft_golden.npz")["tone2048"]
    y = np.empty_like(x)
    _fft(gpu, 2048, gpu.CLFFT_FORWARD).work(1, [x], [y])
    assert abs(y[2047] - 2048j) < 1e-2
    assert np.abs(np.delete(y, 2047)).max() < 1e-2


def test_golden_vectors(gpu):
    g = golden("fft_golden.npz")
    for n in (8, 64, 1024, 4096):
        x = g["x%d" % n]
        y = np.empty_like(x)
        _fft(gpu, n, gpu.CLFFT_FORWARD).work(2, [x], [y])
        assert relerr(y, g["fwd%d" % n]) <= TOL
        _fft(gpu, n, gpu.CLFFT_BACKWARD).work(2, [x], [y])
        assert relerr(y, g["inv%d" % n]) <= TOL
    x = g["x4096"]
    y = np.empty_like(x)
    _fft(gpu, 4096, gpu.CLFFT_FORWARD, g["blackman4096"], shift=True).work(2, [x], [y])
    assert relerr(y, g["fwd_win_shift4096"]) <= TOL
    _fft(gpu, 4096, gpu.CLFFT_BACKWARD, shift=True).work(2, [x], [y])
    assert relerr(y, g["inv_shift4096"]) <= TOL
    xr = g["xr1024"]
    yr = np.empty(xr.size, np.complex64)
    _fft(gpu, 1024, gpu.CLFFT_FORWARD, dtype=gpu.DTYPE_FLOAT).work(2, [xr], [yr])
    assert relerr(yr, g["fwd_real1024"]) <= TOL


@pytest.mark.parametrize("n", [2, 8, 16, 32, 64, 256, 4096, 8192, 16384])  # N <= 64: LDS-redistributed path; > 4096: sub-transform kernel
@pytest.mark.parametrize("fwd,shift,win", [(True, True, True), (True, False, True), (False, True, True), (False, True, False),
                                           (True, True, False)])
def test_window_shift_matrix_vs_oracle(gpu, oracle, n, fwd, shift, win):
    rng = np.random.default_rng(n + 7)
    w = oracle.window(oracle.WIN_BLACKMAN_HARRIS, n) if win else None  # GRC default window
    nvec = 5 if n > 64 else 4096 // n + 3  # more than one workgroup pass, ragged
    x = crandn(rng, nvec * n)
    y = np.empty_like(x)
    _fft(gpu, n, gpu.CLFFT_FORWARD if fwd else gpu.CLFFT_BACKWARD, w, shift=shift).work(nvec, [x], [y])
    assert relerr(y, oracle.fft_block(n, fwd, w, shift, oracle.DTYPE_COMPLEX, x, f64=True)) <= TOL


@pytest.mark.parametrize("n", [4, 16, 64, 8192, 16384])
def test_real_input_small_and_large_sizes(gpu, oracle, n):
    rng = np.random.default_rng(n + 11)
    nvec = 4096 // n + 5
    x = rng.standard_normal(nvec * n).astype(np.float32)
    y = np.empty(nvec * n, np.complex64)
    w = oracle.window(oracle.WIN_HANN, n)
    _fft(gpu, n, gpu.CLFFT_FORWARD, w, dtype=gpu.DTYPE_FLOAT, shift=True).work(nvec, [x], [y])
    assert relerr(y, oracle.fft_block(n, True, w, True, oracle.DTYPE_FLOAT, x, f64=True)) <= TOL


def test_real_input_and_streams(gpu, oracle):
    rng = np.random.default_rng(3)
    n, nvec = 512, 9
    xs = [rng.standard_normal(nvec * n).astype(np.float32) for _ in range(3)]
    ys = [np.empty(nvec * n, np.complex64) for _ in range(3)]
    w = oracle.window(oracle.WIN_HAMMING, n)
    blk = _fft(gpu, n, gpu.CLFFT_FORWARD, w, dtype=gpu.DTYPE_FLOAT, nstreams=3, shift=True)
    blk.work(nvec, xs, ys)
    for x, y in zip(xs, ys):
        assert relerr(y, oracle.fft_block(n, True, w, True, oracle.DTYPE_FLOAT, x, f64=True)) <= TOL


def test_host_path_multi_chunk(gpu, oracle):
    rng = np.random.default_rng(4)
    n, nvec = 4096, 700  # 700 frames * 32 KiB > two 8 MiB staging chunks
    x = crandn(rng, n * nvec)
    y = np.empty_like(x)
    _fft(gpu, n, gpu.CLFFT_FORWARD, shift=True).work(nvec, [x], [y])
    ref = oracle.fft_block(n, True, None, True, oracle.DTYPE_COMPLEX, x[:8 * n], f64=True)
    assert relerr(y[:8 * n], ref) <= TOL
    tail = oracle.fft_block(n, True, None, True, oracle.DTYPE_COMPLEX, x[-3 * n:], f64=True)
    assert relerr(y[-3 * n:], tail) <= TOL
    # Parseval on every frame
    ex = (np.abs(x.reshape(nvec, n)) ** 2).sum(1)
    ey = (np.abs(y.reshape(nvec, n)) ** 2).sum(1) / n
    assert np.allclose(ex, ey, rtol=1e-4)


def test_device_path_full_size_properties(gpu, oracle):
    """BASELINE config 2 at full size (16384 frames of 4096, 1 GiB in+out) on torch's stream:
    round trip ifft(fft(x)) = N*x, Parseval, and a sampled oracle check."""
    import torch
    n, nvec = 4096, 16384
    g = torch.Generator(device="cuda").manual_seed(1234)
    x = torch.randn(nvec * n, 2, device="cuda", generator=g)
    y = torch.empty_like(x)
    z = torch.empty_like(x)
    w = oracle.window(oracle.WIN_BLACKMAN, n)
    fwd = _fft(gpu, n, gpu.CLFFT_FORWARD, shift=True)
    inv = _fft(gpu, n, gpu.CLFFT_BACKWARD, shift=True)
    fwd.work_device(nvec, [x], [y])
    inv.work_device(nvec, [y], [z])
    torch.cuda.synchronize()
    assert torch.allclose(z, x * n, rtol=0, atol=2e-5 * n * 6)
    ex = (x * x).sum().item()
    ey = (y * y).sum().item() / n
    assert abs(ex - ey) / ex < 1e-5
    fw = _fft(gpu, n, gpu.CLFFT_FORWARD, w, shift=True)
    fw.work_device(nvec, [x], [y])
    torch.cuda.synchronize()
    for f0 in (0, 7777, nvec - 2):
        xs = x[f0 * n:(f0 + 2) * n].cpu().numpy().view(np.complex64).reshape(-1)
        ys = y[f0 * n:(f0 + 2) * n].cpu().numpy().view(np.complex64).reshape(-1)
        assert relerr(ys, oracle.fft_block(n, True, w, True, oracle.DTYPE_COMPLEX, xs, f64=True)) <= TOL


def test_linearity_and_impulse(gpu):
    n = 1024
    rng = np.random.default_rng(8)
    a, b = crandn(rng, n), crandn(rng, n)
    blk = _fft(gpu, n, gpu.CLFFT_FORWARD)
    ya, yb, yab = (np.empty(n, np.complex64) for _ in range(3))
    blk.work(1, [a], [ya]); blk.work(1, [b], [yb]); blk.work(1, [(2 * a + 3j * b).astype(np.complex64)], [yab])
    assert relerr(yab, 2 * ya + 3j * yb) <= TOL
    d = np.zeros(n, np.complex64); d[5] = 1
    blk.work(1, [d], [ya])
    assert relerr(ya, np.exp(-2j * np.pi * 5 * np.arange(n) / n)) <= TOL


def test_zero_vectors_and_bad_sizes(gpu):
    blk = _fft(gpu, 64, gpu.CLFFT_FORWARD)
    e = np.empty(0, np.complex64)
    assert blk.work(0, [e], [e]) == 0
    with pytest.raises(gpu.Mi355Error):
        _fft(gpu, 1 << 25, gpu.CLFFT_FORWARD)  # powers of two above 2^24 (the reference's clFFT limit for single precision) are refused, not emulated
    with pytest.raises(gpu.Mi355Error):
        _fft(gpu, (1 << 23) + 1, gpu.CLFFT_FORWARD)  # other sizes above 2^23 too (their chirp-z transform would exceed 2^24 points)
    with pytest.raises(gpu.Mi355Error):
        _fft(gpu, 1, gpu.CLFFT_FORWARD)


# sizes that are not a power of two (clFFT's radix-3/5/7 plans in the reference): lengths 2^a 3^b 5^c 7^d 11^e 13^f up to 15360 run through the
# mixed-radix kernel (fft_mr.hip), everything else -- and every length when MI355_FFT_NO_MR is set -- through the chirp-z path
@pytest.mark.parametrize("n", [3, 5, 12, 48, 100, 1000, 1536, 2000, 4095, 6000, 8191, 10000, 16383])
@pytest.mark.parametrize("fwd,shift,win", [(True, False, False), (True, True, True), (False, True, True), (False, False, False)])
@pytest.mark.parametrize("chirpz_only", [False, True])
def test_sizes_that_are_not_a_power_of_two(gpu, oracle, monkeypatch, n, fwd, shift, win, chirpz_only):
    if chirpz_only:
        if n in (3, 5, 8191, 16383):
            pytest.skip("not a 2-3-5-7-11-13 length (or too short): the chirp-z path either way")
        monkeypatch.setenv("MI355_FFT_NO_MR", "1")  # read when the block is made
    rng = np.random.default_rng(n + 3)
    nvec = 3 if n > 2048 else 7
    w = oracle.window(oracle.WIN_HAMMING, n) if win else None
    x = crandn(rng, nvec * n)
    y = np.empty_like(x)
    blk = _fft(gpu, n, gpu.CLFFT_FORWARD if fwd else gpu.CLFFT_BACKWARD, w, shift=shift)
    assert blk.work(nvec, [x], [y]) == nvec
    assert relerr(y, oracle.fft_block(n, fwd, w, shift, oracle.DTYPE_COMPLEX, x, f64=True)) <= TOL


# the mixed-radix kernel: every radix (2 ... 16 incl. the primes 11, 13 and the composite 6, 9, 10, 12, 14, 15), odd lengths (shift by floor / ceil of n / 2), the longest lengths per workgroup size
# (256 threads: 3584 with a factor 7, else 3840; 512: 7168 / 7680; 1024: 14336 / 15360), ragged frame counts, real input, device path
MR_SIZES = [6, 14, 15, 21, 35, 56, 81, 96, 105, 112, 144, 196, 210, 360, 675, 729, 1125, 1715, 1728, 2401, 2744, 3375, 3584, 3840, 4000, 4200,
            5000, 5625, 7168, 7680, 9000, 10000, 10240 - 10, 12000, 12005, 14336, 15000, 15360,  # (81 = 9 x 9, 144 = 12 x 12, 196 = 14 x 14, 10000 = 10^4)
            22, 143, 1100, 1331, 2197, 2860, 11264, 13312]  # radices 11 and 13 (11 / 13 values per thread: 11264 and 13312 are the longest)


@pytest.mark.parametrize("n", MR_SIZES)
def test_mixed_radix_lengths(gpu, oracle, n):
    import torch
    rng = np.random.default_rng(n)
    nvec = 11 if n < 2000 else 3
    w = oracle.window(oracle.WIN_BLACKMAN, n)
    x = crandn(rng, nvec * n)
    y = np.empty_like(x)
    ref_fn = (lambda f, ww, sh, xx: oracle.fft_block(n, f, ww, sh, oracle.DTYPE_COMPLEX, xx, f64=True)) if n <= 2401 else \
             (lambda f, ww, sh, xx: _np_fft_block(n, f, ww, sh, xx))
    for fwd, shift, win in ((True, True, True), (False, True, True), (True, False, False), (False, False, False)):
        blk = _fft(gpu, n, gpu.CLFFT_FORWARD if fwd else gpu.CLFFT_BACKWARD, w if win else None, shift=shift)
        assert blk.work(nvec, [x], [y]) == nvec
        err = relerr(y, ref_fn(fwd, w if win else None, shift, x))
        assert err <= TOL and err <= 3e-6, (n, fwd, shift, err)
    # real input, forward + shift, on device buffers
    xr = rng.standard_normal(nvec * n).astype(np.float32)
    blk = _fft(gpu, n, gpu.CLFFT_FORWARD, w, dtype=gpu.DTYPE_FLOAT, shift=True)
    dx = torch.from_numpy(xr).cuda()
    dy = torch.empty(nvec * n, 2, device="cuda")
    blk.work_device(nvec, [dx], [dy])
    torch.cuda.synchronize()
    assert relerr(dy.cpu().numpy().view(np.complex64).reshape(-1), ref_fn(True, w, True, xr.astype(np.complex64))) <= TOL


def test_mixed_radix_many_frames_and_tone(gpu, oracle):
    # more frame groups than workgroups (grid-stride), a ragged last group, and a closed form: one cycle-k tone -> n in bin k
    n, nvec = 1000, 20011
    rng = np.random.default_rng(5)
    x = crandn(rng, nvec * n)
    y = np.empty_like(x)
    _fft(gpu, n, gpu.CLFFT_FORWARD).work(nvec, [x], [y])
    for f0 in (0, 777, nvec - 3):
        sl = slice(f0 * n, (f0 + 3) * n)
        assert relerr(y[sl], oracle.fft_block(n, True, None, False, oracle.DTYPE_COMPLEX, x[sl], f64=True)) <= TOL
    n = 12000
    t = np.exp(2j * np.pi * 4321 * np.arange(n) / n).astype(np.complex64)
    z = np.empty_like(t)
    _fft(gpu, n, gpu.CLFFT_FORWARD).work(1, [t], [z])
    assert abs(z[4321] - n) < 0.05 and np.abs(np.delete(z, 4321)).max() < 0.05


def test_chirpz_real_input_device_path_and_chunks(gpu, oracle):
    import torch
    n, nvec = 1200, 40
    rng = np.random.default_rng(8)
    x = rng.standard_normal(nvec * n).astype(np.float32)
    w = oracle.window(oracle.WIN_BLACKMAN, n)
    ref = oracle.fft_block(n, True, w, True, oracle.DTYPE_FLOAT, x, f64=True)
    blk = _fft(gpu, n, gpu.CLFFT_FORWARD, w, dtype=gpu.DTYPE_FLOAT, shift=True)
    y = np.empty(nvec * n, np.complex64)
    blk.work(nvec, [x], [y])
    assert relerr(y, ref) <= TOL
    dx = torch.from_numpy(x).cuda()
    dy = torch.empty(nvec * n, 2, device="cuda")
    blk.work_device(nvec, [dx], [dy])
    torch.cuda.synchronize()
    assert relerr(dy.cpu().numpy().view(np.complex64).reshape(-1), ref) <= TOL
    # a call larger than one work-buffer chunk (128 MiB / (4096 * 8 B) = 4096 frames): first and last frames
    nbig = 5000
    xb = crandn(rng, nbig * n)
    yb = np.empty_like(xb)
    blk2 = _fft(gpu, n, gpu.CLFFT_FORWARD)
    blk2.work(nbig, [xb], [yb])
    for sl in (slice(0, 2 * n), slice((nbig - 2) * n, nbig * n)):
        assert relerr(yb[sl], oracle.fft_block(n, True, None, False, oracle.DTYPE_COMPLEX, xb[sl], f64=True)) <= TOL



# 32768 / 65536: two kernels through a workspace (sub-transforms + radix-8/16 combine)
@pytest.mark.parametrize("n", [32768, 65536])
@pytest.mark.parametrize("fwd,shift,win", [(True, False, False), (True, True, True), (False, True, True), (False, False, False)])
def test_two_kernel_sizes(gpu, oracle, n, fwd, shift, win):
    rng = np.random.default_rng(n + 5)
    nvec = 3
    w = oracle.window(oracle.WIN_BLACKMAN_HARRIS, n) if win else None
    x = crandn(rng, nvec * n)
    y = np.empty_like(x)
    blk = _fft(gpu, n, gpu.CLFFT_FORWARD if fwd else gpu.CLFFT_BACKWARD, w, shift=shift)
    assert blk.work(nvec, [x], [y]) == nvec
    assert relerr(y, oracle.fft_block(n, fwd, w, shift, oracle.DTYPE_COMPLEX, x, f64=True)) <= TOL


# 131072 .. 1048576: three passes (sub-transforms, radix-16 combine into 65536-point transforms, radix-2/4/8/16 combine)
@pytest.mark.parametrize("n", [131072, 262144, 524288, 1048576])
@pytest.mark.parametrize("fwd,shift,win", [(True, True, True), (False, True, True), (True, False, False)])
def test_sizes_above_65536(gpu, oracle, n, fwd, shift, win):
    rng = np.random.default_rng(n + 9)
    nvec = 2
    w = oracle.window(oracle.WIN_BLACKMAN_HARRIS, n) if win else None
    x = crandn(rng, nvec * n)
    y = np.empty_like(x)
    blk = _fft(gpu, n, gpu.CLFFT_FORWARD if fwd else gpu.CLFFT_BACKWARD, w, shift=shift)
    assert blk.work(nvec, [x], [y]) == nvec
    assert relerr(y, oracle.fft_block(n, fwd, w, shift, oracle.DTYPE_COMPLEX, x, f64=True)) <= TOL
    if fwd and not win:  # one bin, real input, and many frames through the device path (more than one workspace chunk at 2^20 points)
        import torch
        t = np.exp(2j * np.pi * 54321 * np.arange(n) / n).astype(np.complex64)
        z = np.empty_like(t)
        blk.work(1, [t], [z])
        assert abs(z[54321] - n) < 1e-5 * n * 8 and np.abs(np.delete(z, 54321)).max() < 1e-5 * n * 8
        xr = rng.standard_normal(n).astype(np.float32)
        yr = np.empty(n, np.complex64)
        _fft(gpu, n, gpu.CLFFT_FORWARD, dtype=gpu.DTYPE_FLOAT, shift=True).work(1, [xr], [yr])
        assert relerr(yr, oracle.fft_block(n, True, None, True, oracle.DTYPE_FLOAT, xr, f64=True)) <= TOL
        frames = (256 << 20) // (n * 8) + 3
        xd = torch.randn(frames * n, 2, device="cuda")
        yd = torch.empty_like(xd)
        blk.work_device(frames, [xd], [yd])
        for f in (0, frames - 1):
            xs = xd[f * n:(f + 1) * n].cpu().numpy().view(np.complex64).reshape(-1)
            ys = yd[f * n:(f + 1) * n].cpu().numpy().view(np.complex64).reshape(-1)
            assert relerr(ys, oracle.fft_block(n, True, None, False, oracle.DTYPE_COMPLEX, xs, f64=True)) <= TOL


# 2^21 .. 2^24 (clFFT's single-precision limit): four passes (sub-transforms, two radix-16 combines, a radix-2/4/8/16 combine)
@pytest.mark.parametrize("n,fwd,shift,win", [(1 << 21, True, True, True), (1 << 22, False, True, True), (1 << 23, True, False, False),
                                             (1 << 24, True, True, True), (1 << 24, False, False, False)])
def test_sizes_above_two_to_the_twenty(gpu, oracle, n, fwd, shift, win):
    import torch
    rng = np.random.default_rng(n % 1000 + 17)
    w = oracle.window(oracle.WIN_HAMMING, n) if win else None
    x = crandn(rng, n)
    blk = _fft(gpu, n, gpu.CLFFT_FORWARD if fwd else gpu.CLFFT_BACKWARD, w, shift=shift)
    xd = torch.from_numpy(x.view(np.float32).reshape(-1, 2)).cuda()
    yd = torch.empty_like(xd)
    blk.work_device(1, [xd], [yd])
    torch.cuda.synchronize()
    y = yd.cpu().numpy().view(np.complex64).reshape(-1)
    assert relerr(y, oracle.fft_block(n, fwd, w, shift, oracle.DTYPE_COMPLEX, x, f64=True)) <= TOL
    if n == 1 << 21:  # the host path (one frame per staging chunk) and a single tone
        y2 = np.empty_like(x)
        assert blk.work(1, [x], [y2]) == 1
        assert np.array_equal(y2, y)
        t = np.exp(2j * np.pi * 1234567 * np.arange(n) / n).astype(np.complex64)
        z = np.empty_like(t)
        _fft(gpu, n, gpu.CLFFT_FORWARD).work(1, [t], [z])
        assert abs(z[1234567] - n) < 1e-5 * n * 8 and np.abs(np.delete(z, 1234567)).max() < 1e-5 * n * 8


def _np_fft_block(n, fwd, w, shift, x):
    """clFFT work() semantics (oracle/o_fft.c:140-188) on numpy's float64 pocketfft: the oracle's O(N^2) DFT for lengths that are not a
    power of two cannot be run at 10^5 .. 10^7 points.  Tied to the oracle at a small length in the test below."""
    x = x.astype(np.complex128).reshape(-1, n)
    if w is not None:
        x = x * np.asarray(w, np.float64)
    if not fwd and shift:
        half = n // 2
        x = np.concatenate([x[:, half:], x[:, :half]], axis=1)  # original position i -> i + (n - half) for i < half
    y = np.fft.fft(x, axis=1) if fwd else np.fft.ifft(x, axis=1) * n
    if fwd and shift:
        ln = (n + 1) // 2
        y = np.concatenate([y[:, ln:], y[:, :ln]], axis=1)
    return y.reshape(-1).astype(np.complex64)


# lengths above 16384 that are not a power of two: chirp-z over the multi-pass power-of-two sizes (65536 .. 2^24 points)
@pytest.mark.parametrize("n,fwd,shift,win", [(16385, True, True, True), (20000, False, True, True), (65537, True, False, False), (100000, True, True, True),
                                             (1000003, True, False, False), (3000000, False, True, False), (8388608 - 1, True, True, False)])
def test_long_sizes_that_are_not_a_power_of_two(gpu, oracle, n, fwd, shift, win):
    rng = np.random.default_rng(n % 977)
    for m, f, sh in ((1000, True, True), (1001, False, True)):  # the numpy reference against the oracle where the oracle can run
        xs = crandn(rng, m)
        ws = oracle.window(oracle.WIN_HAMMING, m)
        assert relerr(_np_fft_block(m, f, ws, sh, xs), oracle.fft_block(m, f, ws, sh, oracle.DTYPE_COMPLEX, xs, f64=True)) <= 1e-6
    w = oracle.window(oracle.WIN_HAMMING, n) if win else None
    x = crandn(rng, n)
    y = np.empty_like(x)
    blk = _fft(gpu, n, gpu.CLFFT_FORWARD if fwd else gpu.CLFFT_BACKWARD, w, shift=shift)
    assert blk.work(1, [x], [y]) == 1
    assert relerr(y, _np_fft_block(n, fwd, w, shift, x)) <= TOL
    if n == 20000:  # real input and two frames
        xr = rng.standard_normal(2 * n).astype(np.float32)
        yr = np.empty(2 * n, np.complex64)
        _fft(gpu, n, gpu.CLFFT_FORWARD, dtype=gpu.DTYPE_FLOAT).work(2, [xr], [yr])
        assert relerr(yr, _np_fft_block(n, True, None, False, xr.astype(np.complex64))) <= TOL


# 2-3-5-7-11-13 lengths above 15360 points: two passes with the mixed-radix passes inside (k_fft_mr_tile); row lengths that are / are not a
# multiple of sixteen values, an odd length (shift by floor / ceil), the longest length (960 x 960), real input, several frames
@pytest.mark.parametrize("n,nvec", [(16000, 5), (20000, 3), (22050, 2), (30030, 2), (44100, 2), (48000, 3), (50625, 2), (100000, 2), (250000, 1),
                                    (921600, 1)])
def test_mixed_radix_two_pass_lengths(gpu, oracle, n, nvec):
    import torch
    rng = np.random.default_rng(n % 1009)
    w = oracle.window(oracle.WIN_HAMMING, n)
    x = crandn(rng, nvec * n)
    y = np.empty_like(x)
    for fwd, shift, win in ((True, True, True), (False, True, True), (True, False, False), (False, False, False)):
        blk = _fft(gpu, n, gpu.CLFFT_FORWARD if fwd else gpu.CLFFT_BACKWARD, w if win else None, shift=shift)
        assert blk.work(nvec, [x], [y]) == nvec
        err = relerr(y, _np_fft_block(n, fwd, w if win else None, shift, x))
        assert err <= TOL and err <= 3e-6, (n, fwd, shift, err)
    xr = rng.standard_normal(nvec * n).astype(np.float32)
    blk = _fft(gpu, n, gpu.CLFFT_FORWARD, w, dtype=gpu.DTYPE_FLOAT, shift=True)
    dx = torch.from_numpy(xr).cuda()
    dy = torch.empty(nvec * n, 2, device="cuda")
    blk.work_device(nvec, [dx], [dy])
    torch.cuda.synchronize()
    assert relerr(dy.cpu().numpy().view(np.complex64).reshape(-1), _np_fft_block(n, True, w, True, xr.astype(np.complex64))) <= TOL


def test_two_pass_length_through_chirpz_too(gpu, oracle, monkeypatch):
    monkeypatch.setenv("MI355_FFT_NO_MR", "1")  # the chirp-z path keeps its coverage at a length the mixed-radix form now takes
    n = 20000
    rng = np.random.default_rng(3)
    x = crandn(rng, 2 * n)
    y = np.empty_like(x)
    _fft(gpu, n, gpu.CLFFT_FORWARD, shift=True).work(2, [x], [y])
    assert relerr(y, _np_fft_block(n, True, None, True, x)) <= TOL


def test_two_kernel_real_input_and_tone(gpu, oracle):
    n = 32768
    rng = np.random.default_rng(2)
    x = rng.standard_normal(2 * n).astype(np.float32)
    y = np.empty(2 * n, np.complex64)
    _fft(gpu, n, gpu.CLFFT_FORWARD, dtype=gpu.DTYPE_FLOAT, shift=True).work(2, [x], [y])
    assert relerr(y, oracle.fft_block(n, True, None, True, oracle.DTYPE_FLOAT, x, f64=True)) <= TOL
    t = np.exp(2j * np.pi * 12345 * np.arange(65536) / 65536).astype(np.complex64)  # one bin
    z = np.empty_like(t)
    _fft(gpu, 65536, gpu.CLFFT_FORWARD).work(1, [t], [z])
    assert abs(z[12345] - 65536) < 1.0 and np.abs(np.delete(z, 12345)).max() < 1.0


def test_reference_cli_tone_through_window_and_shift(gpu):
    """The input the reference's clFFT timing CLI builds (lib/test_clenabled.cc:835-851): (sin, cos) of one cycle per frame in float
    arithmetic; unwindowed its spectrum is j*N in bin N-1 (closed form), windowed + shifted it is compared with the float64 fixture."""
    import json
    import os
    from conftest import GOLDEN
    g = golden("cli_golden.npz")
    with open(os.path.join(GOLDEN, "cli_kat.json")) as f:
        k = json.load(f)["fft_tone_4096"]
    x = np.tile(g["tone4096_x"], 3)
    y = np.empty_like(x)
    blk = gpu.clFFT(4096, gpu.CLFFT_FORWARD, None, gpu.DTYPE_COMPLEX, *GPU_ARGS, 0, 1, False)
    assert blk.testOpenCL(x.size, [x], [y]) == x.size  # the CLI's hook counts samples (lib/clFFT_impl.cc:520-524)
    for v in range(3):
        X = y[v * 4096:(v + 1) * 4096]
        assert abs(X[k["peak_bin_unshifted"]] - complex(*k["peak"])) < 1e-2
        assert np.abs(np.delete(X, k["peak_bin_unshifted"])).max() < 2e-2
        assert relerr(X, g["tone4096_fwd"]) <= TOL
    blk = gpu.clFFT(4096, gpu.CLFFT_FORWARD, g["tone4096_win"], gpu.DTYPE_COMPLEX, *GPU_ARGS, 0, 1, True)
    blk.work(3, [x], [y])
    for v in range(3):
        assert relerr(y[v * 4096:(v + 1) * 4096], g["tone4096_fwd_win_shift"]) <= TOL


@pytest.mark.parametrize("case,n", [("ta", 4096), ("tb", 1000), ("tc", 4099), ("td", 64)])
def test_independent_scipy_cases(gpu, case, n):
    """Against scipy.fft (pocketfft) + scipy.signal.windows.blackman -- an implementation that is not this repository's
    (tests/golden/gen_golden.py::independent_golden): BASELINE config 2's length, a 2^3 5^3 length (mixed-radix kernel), a prime (chirp-z)
    and a small one; windowed + shifted forward, plain forward, unscaled inverse."""
    g = golden("independent_golden.npz")
    x, w = g[case + "_x"], g[case + "_win"]
    nvec = x.size // n
    y = np.empty_like(x)
    assert _fft(gpu, n, gpu.CLFFT_FORWARD, window=w, shift=True).work(nvec, [x], [y]) == nvec
    assert relerr(y, g[case + "_fwd_win_shift"]) <= TOL
    _fft(gpu, n, gpu.CLFFT_FORWARD).work(nvec, [x], [y])
    assert relerr(y, g[case + "_fwd"]) <= TOL
    _fft(gpu, n, gpu.CLFFT_BACKWARD).work(nvec, [x], [y])
    assert relerr(y, g[case + "_inv"]) <= TOL
