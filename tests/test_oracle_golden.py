"""CPU: pins the oracle against the reference's known answers, the values the
survey recorded from the reference's own files, and float64 definitions."""
import json
import os

import numpy as np

from conftest import GOLDEN, crandn, golden, relerr


def _kat():
    with open(os.path.join(GOLDEN, "kat.json")) as f:
        return json.load(f)


def test_mathop_known_answers(oracle):
    k = _kat()["mathop_multiply"]
    a = np.full(k["n"], complex(*k["a"]), np.complex64)
    out = oracle.mathop(oracle.DTYPE_COMPLEX, oracle.OP_MULTIPLY, a, a)
    assert np.all(out == np.complex64(complex(*k["expect"])))
    k = _kat()["mathconst_multiply"]
    out = oracle.mathconst(oracle.DTYPE_COMPLEX, oracle.OP_MULTIPLY, k["k"], np.full(64, complex(*k["a"]), np.complex64))
    assert np.all(out == np.complex64(complex(*k["expect"])))


def test_mathop_all_ops_vs_numpy(oracle):
    rng = np.random.default_rng(1)
    a, b = crandn(rng, 1001), crandn(rng, 1001)
    o = oracle
    assert relerr(o.mathop(1, o.OP_MULTIPLY, a, b), a.astype(np.complex128) * b) < 1e-6
    assert np.array_equal(o.mathop(1, o.OP_ADD, a, b), a + b)
    assert np.array_equal(o.mathop(1, o.OP_SUBTRACT, a, b), a - b)
    assert relerr(o.mathop(1, o.OP_MULTIPLY_CONJUGATE, a, b), a.astype(np.complex128) * np.conj(b)) < 1e-6
    assert np.array_equal(o.mathconst(1, o.OP_CONJUGATE, 0, a), np.conj(a))
    # the REAL constant is added to both components (lib/clMathConst_impl.cc:194-197)
    assert np.array_equal(o.mathconst(1, o.OP_ADD, 2.5, a), (a.real + np.float32(2.5)) + 1j * (a.imag + np.float32(2.5)))
    ia = rng.integers(-2**31, 2**31, 777, dtype=np.int64).astype(np.int32)
    ib = rng.integers(-2**31, 2**31, 777, dtype=np.int64).astype(np.int32)
    assert np.array_equal(o.mathop(3, o.OP_MULTIPLY, ia, ib), (ia.astype(np.int64) * ib).astype(np.int32))  # wraps
    assert np.array_equal(o.mathop(3, o.OP_ADD, ia, ib), (ia.astype(np.int64) + ib).astype(np.int32))
    assert np.array_equal(o.mathconst(3, o.OP_MULTIPLY, 3.0, ia), (ia.astype(np.int64) * 3).astype(np.int32))
    fa, fb = a.real.copy(), b.real.copy()
    assert np.array_equal(o.mathop(2, o.OP_MULTIPLY, fa, fb), fa * fb)


def test_window_and_firdes_recorded_values(oracle):
    k = _kat()
    w = oracle.window(oracle.WIN_BLACKMAN, 4096)
    assert np.float32(w[1]) == np.float32(k["window_blackman_4096"]["i1"])
    assert np.float32(w[2048]) == np.float32(k["window_blackman_4096"]["i2048"])
    t = oracle.firdes_low_pass(*k["firdes_low_pass_65"]["args"])
    assert t.size == 65 and t[0] == np.float32(k["firdes_low_pass_65"]["t0"]) and t[32] == np.float32(k["firdes_low_pass_65"]["t32"])
    t = oracle.firdes_low_pass(*k["firdes_low_pass_2047"]["args"])
    assert t.size == 2047 and t[0] == np.float32(k["firdes_low_pass_2047"]["t0"])
    assert t[1023] == np.float32(k["firdes_low_pass_2047"]["t1023"])
    assert oracle.firdes_low_pass(*k["firdes_low_pass_145"]["args"]).size == 145
    g = golden("filter_golden.npz")
    assert relerr(oracle.firdes_low_pass(1.0, 10e6, 1e6, 372000.0), g["taps65"]) < 1e-6
    assert relerr(w, golden("fft_golden.npz")["blackman4096"]) < 1e-6
    # symmetric linear-phase design, unit DC gain
    t = oracle.firdes_low_pass(1.0, 10e6, 1e6, 372000.0)
    assert np.allclose(t, t[::-1], atol=1e-7) and abs(float(t.sum()) - 1.0) < 1e-6


def test_windows_vs_scipy(oracle):
    """Independent implementation (scipy.signal.windows, symmetric form = GNU Radio's N-1 denominators) for every window type
    the reference's window.cc builds.  Flat-top: the reference's last coefficient is 0.028/4.63867 (lib/window.cc:243-248), scipy's
    0.0322/4.63867 -- compared against the cosine sum with the reference's coefficients, and against scipy within that difference."""
    from scipy.signal import windows as W
    o = oracle
    for n in (2, 7, 64, 65, 1000, 4096):
        pairs = [(o.WIN_HAMMING, W.hamming(n, sym=True)), (o.WIN_HANN, W.hann(n, sym=True)), (o.WIN_BLACKMAN, W.blackman(n, sym=True)),
                 (o.WIN_RECTANGULAR, np.ones(n)), (o.WIN_BLACKMAN_HARRIS, W.blackmanharris(n, sym=True)),
                 (o.WIN_BARTLETT, W.bartlett(n, sym=True))]
        for wt, ref in pairs:
            assert np.abs(o.window(wt, n).astype(np.float64) - ref).max() < 3e-7, (wt, n)
        for beta in (0.0, 3.5, 6.76, 12.0):
            assert np.abs(o.window(o.WIN_KAISER, n, beta).astype(np.float64) - W.kaiser(n, beta, sym=True)).max() < 3e-7, (n, beta)
        k = np.arange(n)
        c = np.array([1.0, 1.93, 1.29, 0.388, 0.028]) / 4.63867
        ft = sum(((-1) ** i) * c[i] * np.cos(2 * np.pi * i * k / max(n - 1, 1)) for i in range(5))
        assert np.abs(o.window(o.WIN_FLATTOP, n).astype(np.float64) - ft).max() < 3e-7
        assert np.abs(o.window(o.WIN_FLATTOP, n).astype(np.float64) - W.flattop(n, sym=True)).max() < 2.1 * (0.0322 - 0.028) / 4.63867


def test_firdes_low_pass_vs_scipy_firwin(oracle):
    """scipy.signal.firwin = windowed sinc scaled to unit DC gain: an independent design of the three fixture filters (and the
    other window types), within float32 rounding of the oracle's restatement of firdes::low_pass."""
    from scipy.signal import firwin
    o = oracle
    for args in ((1.0, 10e6, 1e6, 372000.0), (1.0, 64.0, 0.5, 0.0753), (1.0, 300e3, 48e3, 5e3), (2.5, 48000.0, 3000.0, 900.0)):
        gain, fs, fc, tw = args
        t = o.firdes_low_pass(*args)
        ref = gain * firwin(t.size, fc, window="hamming", fs=fs, scale=True)
        assert np.abs(t.astype(np.float64) - ref).max() < 2e-7 * max(1.0, gain), args
    for wt, name in ((o.WIN_HANN, "hann"), (o.WIN_BLACKMAN, "blackman"), (o.WIN_BLACKMAN_HARRIS, "blackmanharris"), (o.WIN_RECTANGULAR, "boxcar")):
        t = o.firdes_low_pass(1.0, 32000.0, 4000.0, 1000.0, wt)
        assert np.abs(t.astype(np.float64) - firwin(t.size, 4000.0, window=name, fs=32000.0, scale=True)).max() < 2e-7, name
    t = o.firdes_low_pass(1.0, 32000.0, 4000.0, 1000.0, o.WIN_KAISER, 6.76)
    assert np.abs(t.astype(np.float64) - firwin(t.size, 4000.0, window=("kaiser", 6.76), fs=32000.0, scale=True)).max() < 2e-7


def test_reference_cli_closed_forms(oracle):
    """The inputs the reference's timing CLIs build have closed-form answers (tests/golden/cli_kat.json, cli_golden.npz):
    ramp taps i/1000 over a constant (1, 0.5) stream (lib/test-clfilter.cc:76-80,98-100) and the (sin, cos) tone of
    lib/test_clenabled.cc:835-851 through window + shift."""
    with open(os.path.join(GOLDEN, "cli_kat.json")) as f:
        k = json.load(f)
    for case in k["filter_ramp_taps"]:
        nt = case["ntaps"]
        taps = (np.arange(nt, dtype=np.float32) / np.float32(1000.0)).astype(np.float32)
        x = np.full(4096 + nt - 1, complex(*case["input"]), np.complex64)
        want = complex(*case["expect"])
        y = oracle.fir_ccf(taps, x, 4096)
        assert np.abs(y - want).max() <= 1e-5 * abs(want), nt
        f = oracle.FFTFilter(1, taps)
        yf = f.filter(4096, x[:4096])
        whole = (4096 // f.nsamples) * f.nsamples  # the stateful filter works in blocks of nsamples; after its start-up transient
        assert np.abs(yf[nt:whole] - want).max() <= 1e-5 * abs(want), nt
    g = golden("cli_golden.npz")
    o = oracle
    X = o.fft_block(4096, True, None, False, o.DTYPE_COMPLEX, g["tone4096_x"])
    kk = k["fft_tone_4096"]
    assert abs(X[kk["peak_bin_unshifted"]] - complex(*kk["peak"])) < 1e-2 and np.abs(np.delete(X, kk["peak_bin_unshifted"])).max() < 2e-2
    assert relerr(X, g["tone4096_fwd"]) < 2e-6
    Xw = o.fft_block(4096, True, g["tone4096_win"], True, o.DTYPE_COMPLEX, g["tone4096_x"])
    assert relerr(Xw, g["tone4096_fwd_win_shift"]) < 2e-6


def test_fft_tone_known_answer(oracle):
    k = _kat()["fft_tone"]
    x = golden("fft_golden.npz")["tone2048"]
    for f64 in (False, True):
        X = oracle.fft(x, -1, f64=f64)
        assert abs(X[k["peak_bin"]] - complex(*k["peak"])) < 1e-3
        assert np.abs(np.delete(X, k["peak_bin"])).max() < k["others_abs_max"]


def test_fft_vs_float64(oracle):
    g = golden("fft_golden.npz")
    for n in (8, 64, 1024, 4096):
        x = g["x%d" % n]
        for v in range(2):
            fr = x[v * n:(v + 1) * n]
            assert relerr(oracle.fft(fr, -1), g["fwd%d" % n][v * n:(v + 1) * n]) < 2e-6
            assert relerr(oracle.fft(fr, +1), g["inv%d" % n][v * n:(v + 1) * n]) < 2e-6
            assert relerr(oracle.fft(fr, -1, f64=True), g["fwd%d" % n][v * n:(v + 1) * n]) < 2e-7


def test_fft_block_window_shift(oracle):
    g = golden("fft_golden.npz")
    o = oracle
    out = o.fft_block(4096, True, g["blackman4096"], True, o.DTYPE_COMPLEX, g["x4096"])
    assert relerr(out, g["fwd_win_shift4096"]) < 2e-6
    out = o.fft_block(4096, False, None, True, o.DTYPE_COMPLEX, g["x4096"])
    assert relerr(out, g["inv_shift4096"]) < 2e-6
    out = o.fft_block(1024, True, None, False, o.DTYPE_FLOAT, g["xr1024"])
    assert relerr(out, g["fwd_real1024"]) < 2e-6
    assert o.fft_block(4096, True, None, False, o.DTYPE_COMPLEX, g["x4096"]).size == 8192
    # sizes that are not a power of two: direct DFT in double, checked against numpy's mixed-radix FFT
    rng = np.random.default_rng(12)
    for n in (12, 45, 1000):
        x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
        assert relerr(o.fft(x), np.fft.fft(x.astype(np.complex128)).astype(np.complex64)) < 1e-6


def test_fft_filter_sizes_and_outputs(oracle):
    k = _kat()["fft_filter_sizes_65"]
    g = golden("filter_golden.npz")
    f = oracle.FFTFilter(1, g["taps65"])
    assert (f.fftsize, f.nsamples) == (k["fftsize"], k["nsamples"])
    # tap spectrum pre-scaled by 1/fftsize (lib/fft_filter.cc:52-57)
    H = np.fft.fft(np.concatenate([g["taps65"].astype(np.float64), np.zeros(256 - 65)])) / 256
    assert relerr(f.xformed_taps(), H) < 1e-6
    x = g["x"]
    assert relerr(f.filter(x.size, x), g["y_d1"]) < 2e-6
    for d in (2, 3):
        f = oracle.FFTFilter(d, g["taps65"])
        y = f.filter(x.size // d, x)
        ref = g["y_d%d" % d]
        m = min(y.size, ref.size)
        assert m >= x.size // d and relerr(y[:m], ref[:m]) < 2e-6


def test_fir_vs_fft_filter_and_golden(oracle):
    g = golden("filter_golden.npz")
    x, t = g["x"], g["taps65"]
    xh = np.concatenate([np.zeros(t.size - 1, np.complex64), x])
    y = oracle.fir_ccf(t, xh, x.size)
    assert relerr(y, g["y_d1"]) < 2e-6
    assert relerr(y, oracle.FFTFilter(1, t).filter(x.size, x)) < 2e-6  # the two mirrors agree to ~2e-7
    assert relerr(oracle.fir_ccf(t, xh, x.size // 2, 2), g["y_d2"]) < 2e-6
    assert relerr(oracle.fir_ccf(t, xh, x.size // 3, 3), g["y_d3"]) < 2e-6
    assert relerr(oracle.fir_ccc(g["ctaps65"], xh, x.size), g["yc_d1"]) < 2e-6
    assert relerr(oracle.fir_ccc(g["ctaps65"], xh, x.size // 2, 2), g["yc_d2"]) < 2e-6
    xh7 = np.concatenate([np.zeros(6, np.complex64), x])
    assert relerr(oracle.fir_ccf(g["taps7"], xh7, x.size), g["y7_d1"]) < 2e-6


def test_pfb_vs_closed_form(oracle):
    g = golden("pfb_golden.npz")
    for c in "abc":
        M, R, buf = (int(v) for v in g[c + "_cfg"])
        for f64, tol in ((True, 1e-6), (False, 1e-5)):
            y = oracle.pfb(g[c + "_taps"], buf, M, R, g[c + "_chmap"], g[c + "_x"], f64=f64)
            assert y.size == g[c + "_y"].size
            assert relerr(y, g[c + "_y"]) < tol, (c, f64)
    try:
        oracle.pfb(g["c_taps"], 63, 8, 4, [0], g["c_x"])
        assert False, "buf_items % num_channels != 0 must be refused"
    except ValueError:
        pass


def test_pfb_vs_independent_scipy(oracle):
    """The channelizer restatement against an implementation that is not this repository's: scipy.signal.upfirdn, one channel at a time
    (mix down, FIR, decimate; tests/golden/gen_golden.py::independent_golden) -- the reference flowgraph's 3-channel case, BASELINE
    config 4's shape, a 2-fold and a 4-fold oversampled map."""
    g = golden("independent_golden.npz")
    for c in ("pa", "pb", "pc", "pd"):
        M, R, buf = (int(v) for v in g[c + "_cfg"])
        for f64, tol in ((True, 1e-6), (False, 1e-5)):
            y = oracle.pfb(g[c + "_taps"], buf, M, R, g[c + "_chmap"], g[c + "_x"], f64=f64)
            assert y.size == g[c + "_y"].size and relerr(y, g[c + "_y"]) < tol, (c, f64)


def test_xengine_vs_independent_numpy(oracle):
    """The X-engine restatement against numpy.einsum (integer sums, exact) + numpy.tril_indices (baseline order) and against
    scipy.signal.correlate's zero lag for single baselines: the exact mode bit for bit, the float mode within 1e-5."""
    g = golden("independent_golden.npz")
    kd = 0.007874015748031496063
    for c in ("xa", "xb", "xc"):
        N, F, T, npol = (int(v) for v in g[c + "_cfg"])
        x = g[c + "_x"]
        ref = ((g[c + "_sum_re"].astype(np.float64) * kd * kd).astype(np.float32)
               + 1j * (g[c + "_sum_im"].astype(np.float64) * kd * kd).astype(np.float32)).astype(np.complex64)  # (double)S * kd * kd, one rounding
        assert np.array_equal(oracle.xengine_ichar(N, F, npol, T, x, exact=True), ref), c
        assert relerr(oracle.xengine_ichar(N, F, npol, T, x, exact=False), g[c + "_y"]) < 1e-5, c
        assert relerr(ref, g[c + "_y"]) < 1e-6
    N, F, T, npol = (int(v) for v in g["xa_cfg"])
    v = oracle.xengine_ichar(N, F, npol, T, g["xa_x"], exact=True).reshape(F, N * (N + 1) // 2)
    for (s1, s2, f), want in zip(g["xa_picks"], g["xa_pick_vals"]):
        assert abs(v[f, s1 * (s1 + 1) // 2 + s2] - want) <= 1e-6 * max(1.0, abs(want))


def test_xengine_exact_and_closed_forms(oracle):
    g = golden("xengine_golden.npz")
    N, F, T = (int(v) for v in g["cfg"])
    sc = 0.007874015748031496063 ** 2
    for npol in (1, 2):
        x = g["i8_p%d_x" % npol]
        ref = ((g["i8_p%d_sum_re" % npol] * sc) + 1j * (g["i8_p%d_sum_im" % npol] * sc)).astype(np.complex64)
        assert np.array_equal(oracle.xengine_ichar(N, F, npol, T, x, exact=True), ref)  # bit exact
        assert relerr(oracle.xengine_ichar(N, F, npol, T, x, exact=False), ref) < 1e-5
    assert relerr(oracle.xengine_cf32(N, F, 2, T, g["cf_p2_x"]), g["cf_p2_y"]) < 2e-6
    assert relerr(oracle.xengine_packed4(N, F, T, g["p4_x"]), g["p4_y"]) < 2e-6
    # every sample (127,0) -> V = T for every baseline
    x = np.zeros((T, N, F, 1, 2), np.int8)
    x[..., 0] = 127
    v = oracle.xengine_ichar(N, F, 1, T, x.reshape(-1), exact=False)
    assert np.allclose(v, T, rtol=1e-6)
    # identical unit-modulus tone on every antenna -> T + 0j (lib/test-clxengine.cc:225-247)
    ph = np.exp(2j * np.pi * np.arange(T * F) / 37.0).reshape(T, 1, F).repeat(N, 1).astype(np.complex64)
    v = oracle.xengine_cf32(N, F, 1, T, ph.reshape(-1))
    assert np.allclose(v, T, atol=1e-4)
    # conjugation sense: x_s1 * conj(x_s2), s1 >= s2
    x = np.zeros((1, 2, 1, 1, 2), np.int8)
    x[0, 0, 0, 0] = (127, 0)    # station 0 = 1
    x[0, 1, 0, 0] = (0, 127)    # station 1 = j
    v = oracle.xengine_ichar(2, 1, 1, 1, x.reshape(-1), exact=True)
    assert np.allclose(v, [1, 1j, 1])  # k=0:(0,0) k=1:(1,0)=j*conj(1) k=2:(1,1)
    # pipeline accumulation
    acc = oracle.xengine_ichar(N, F, 1, T, g["i8_p1_x"], exact=True)
    acc2 = oracle.xengine_ichar(N, F, 1, T, g["i8_p1_x"], exact=True, acc=acc.copy())
    assert np.allclose(acc2, 2 * acc)


def test_xengine_gather_layout(oracle):
    o = oracle
    N, F, T = 3, 5, 4
    rng = np.random.default_rng(3)
    ins = [rng.integers(-127, 128, size=T * F * 2, dtype=np.int64).astype(np.int8) for _ in range(2 * N)]
    fb = np.zeros(T * N * F * 2 * 2, np.int8)
    o.xengine_gather(o.DTYPE_BYTE, N, F, 2, T, 0, ins, fb)
    fb = fb.reshape(T, N, F, 2, 2)
    for i in range(N):
        assert np.array_equal(fb[:, i, :, 0, :].reshape(-1), ins[i])
        assert np.array_equal(fb[:, i, :, 1, :].reshape(-1), ins[i + N])
    fb1 = np.zeros(T * N * F * 2, np.int8)
    o.xengine_gather(o.DTYPE_BYTE, N, F, 1, 2, 0, ins[:N], fb1)
    o.xengine_gather(o.DTYPE_BYTE, N, F, 1, 2, 2, [a[2 * F * 2:] for a in ins[:N]], fb1)
    assert np.array_equal(fb1.reshape(T, N, F * 2)[:, 1, :].reshape(-1), ins[1])


def test_widened_rows_vs_golden(oracle):
    """Oracle restatements of the widened rows and of the later FFT sizes vs float64 numpy fixtures."""
    g = golden("widen_golden.npz")
    o = oracle
    ins = [g["xcorr_in%d" % i] for i in range(3)]
    for itype in (1, 2):
        outs = o.xcorr_fft(256, itype, ins, use_f64=True)
        for s in (1, 2):
            assert relerr(outs[s - 1], g["xcorr_t%d_out%d" % (itype, s)]) < 2e-6
    n = g["el_a"].size
    assert relerr(o.elem(1, n, [g["el_a"]], p0=2.5, p1=-3.0)[0], g["el_log10"]) < 1e-6
    assert relerr(o.elem(2, n, [g["el_a"], g["el_b"]], p0=10.0, p1=1.0)[0], g["el_snr"]) < 1e-5
    assert relerr(o.elem(3, n, [g["el_z"][:n]])[0], g["el_mag"]) < 1e-6
    assert relerr(o.elem(4, n, [g["el_z"][:n]])[0], g["el_arg"]) < 1e-6
    assert relerr(o.elem(6, n, [g["el_a"], g["el_ph"]])[0], g["el_mp2c"]) < 1e-6
    assert relerr(o.elem(7, n, [g["el_z"]], p0=0.7)[0], g["el_qdemod"]) < 1e-6
    for nn in (12, 1000):
        y = o.fft_block(nn, True, g["fftw%d" % nn], True, o.DTYPE_COMPLEX, g["fftx%d" % nn], f64=True)
        assert relerr(y, g["fft_fwd_win_shift%d" % nn]) < 1e-6
    assert relerr(o.fft(g["fftx8192"]), g["fft_fwd8192"]) < 2e-6


def test_fft_and_filters_vs_independent_scipy(oracle):
    """clFFT and the filter restatements against library implementations that are not this repository's: scipy.fft (pocketfft) with
    scipy.signal.windows.blackman for the windowed + shifted forward transform, the plain forward and the unscaled inverse (power-of-two
    lengths: the oracle's FFT handles nothing else); scipy.signal.firwin taps through scipy.signal.upfirdn (decimations 1, 2, 3, 10, real and
    complex taps), scipy.signal.lfilter, and scipy.signal.fftconvolve for 3001 taps (tests/golden/gen_golden.py::independent_golden)."""
    g = golden("independent_golden.npz")
    for c, n in (("ta", 4096), ("td", 64)):
        x, w = g[c + "_x"], g[c + "_win"]
        for f64, tol in ((True, 2e-7), (False, 2e-6)):
            assert relerr(oracle.fft_block(n, True, w, True, oracle.DTYPE_COMPLEX, x, f64=f64), g[c + "_fwd_win_shift"]) < tol, (c, f64)
            assert relerr(oracle.fft_block(n, True, None, False, oracle.DTYPE_COMPLEX, x, f64=f64), g[c + "_fwd"]) < tol, (c, f64)
            assert relerr(oracle.fft_block(n, False, None, False, oracle.DTYPE_COMPLEX, x, f64=f64), g[c + "_inv"]) < tol, (c, f64)
    x, taps, ctaps, lt = g["fa_x"], g["fa_taps"], g["fa_ctaps"], g["fa_long_taps"]
    hist = lambda k: np.concatenate([np.zeros(k - 1, np.complex64), x])
    for d in (1, 2, 3, 10):
        n = x.size // d
        assert relerr(oracle.fir_ccf(taps, hist(65), n, d), g["fa_y_d%d" % d]) < 1e-6, d
        assert relerr(oracle.fir_ccc(ctaps, hist(65), n, d), g["fa_yc_d%d" % d]) < 1e-6, d
        f = oracle.FFTFilter(d, taps)            # the stateful overlap-add restatement of fft_filter_ccf
        assert relerr(f.filter(n, x)[:n], g["fa_y_d%d" % d][:n]) < 2e-6, d
    assert relerr(oracle.fir_ccf(taps, hist(65), x.size, 1), g["fa_y_lfilter"]) < 1e-6
    assert relerr(oracle.fir_ccf(lt, hist(3001), x.size, 1), g["fa_y_long"]) < 1e-5  # (float accumulation over 3001 products, as the reference's dot product)


def test_xcorr_vs_independent_scipy(oracle):
    """clxcorrelate_fft_vcf (time-series inputs) against scipy.signal.correlate's direct-form linear correlation folded onto N circular lags."""
    g = golden("independent_golden.npz")
    ins = [g["xc_in%d" % i] for i in range(3)]
    for f64, tol in ((True, 1e-6), (False, 1e-5)):
        for got, s_ in zip(oracle.xcorr_fft(256, 2, ins, use_f64=f64), (1, 2)):
            assert relerr(got, g["xc_out%d" % s_]) < tol, (f64, s_)
