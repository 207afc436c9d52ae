#!/usr/bin/env python3
"""Generates the committed golden fixtures in tests/golden/.

Nothing here imports the oracle or the product: expected outputs come from
(a) known answers the reference's own tests hold, (b) values the survey recorded
from the reference's own source files, and (c) float64
numpy evaluation of the mathematical definitions (numpy.fft, np.convolve,
direct sums).  Run:  python tests/golden/gen_golden.py
"""
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def crandn(rng, n):
    return (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)


def firdes_low_pass64(gain, fs, cutoff, tw, atten=53.0):
    """float64 windowed-sinc (Hamming) low-pass; definition of firdes::low_pass."""
    nt = int(atten * fs / (22.0 * tw))
    nt += (nt & 1) == 0
    M = (nt - 1) // 2
    n = np.arange(-M, M + 1)
    w = 0.54 - 0.46 * np.cos(2 * np.pi * np.arange(nt) / (nt - 1))
    w0 = 2 * np.pi * cutoff / fs
    with np.errstate(invalid="ignore", divide="ignore"):
        t = np.where(n == 0, w0 / np.pi, np.sin(n * w0) / (n * np.pi)) * w
    t = t * (gain / t.sum())
    return t


def widen_golden():
    """Fixtures of the widened rows and of the FFT sizes added later: float64 numpy definitions only."""
    rng = np.random.default_rng(20260928)
    g = {}
    # clxcorrelate_fft_vcf (lib/clxcorrelate_fft_vcf_impl.cc:1058-1143): out = halfswap(|IFFT_unscaled(X0 conj(Xs))|)
    n, nfr = 256, 3
    xs = [crandn(rng, n * nfr) for _ in range(3)]
    for itype in (1, 2):
        X = [x.reshape(nfr, n).astype(np.complex128) for x in xs]
        if itype == 2:
            X = [np.fft.fft(x, axis=1) for x in X]
        for s in (1, 2):
            r = np.fft.ifft(X[0] * np.conj(X[s]), axis=1) * n
            g["xcorr_t%d_out%d" % (itype, s)] = np.fft.fftshift(np.abs(r), axes=1).reshape(-1).astype(np.float32)
    for i, x in enumerate(xs):
        g["xcorr_in%d" % i] = x
    # elementwise family (kernels cited in include/mi355_clenabled.h)
    m = 1000
    a = (np.abs(rng.standard_normal(m)) + 0.05).astype(np.float32)
    b = (np.abs(rng.standard_normal(m)) + 0.05).astype(np.float32)
    z = crandn(rng, m + 1)
    ph = rng.uniform(-10, 10, m).astype(np.float32)
    z64 = z.astype(np.complex128)
    g.update(el_a=a, el_b=b, el_z=z, el_ph=ph,
             el_log10=(2.5 * np.log10(a.astype(np.float64)) - 3.0).astype(np.float32),
             el_snr=np.abs(10 * np.log10(a.astype(np.float64) / b.astype(np.float64)) + 1.0).astype(np.float32),
             el_mag=np.abs(z64[:m]).astype(np.float32), el_arg=np.angle(z64[:m]).astype(np.float32),
             el_mp2c=(a.astype(np.float64) * np.exp(1j * ph.astype(np.float64))).astype(np.complex64),
             el_qdemod=(0.7 * np.angle(z64[1:] * np.conj(z64[:-1]))).astype(np.float32))
    # FFT sizes that are not a power of two (len = ceil(N/2) shift of clFFT_impl::testCPU, :503-507) and the large ones
    for nn in (12, 1000):
        x = crandn(rng, 2 * nn)
        w = (0.54 - 0.46 * np.cos(2 * np.pi * np.arange(nn) / (nn - 1))).astype(np.float32)
        X = np.fft.fft(x.reshape(2, nn).astype(np.complex128) * w, axis=1)
        ln = (nn + 1) // 2
        g["fftx%d" % nn], g["fftw%d" % nn] = x, w
        g["fft_fwd_win_shift%d" % nn] = np.concatenate([X[:, ln:], X[:, :ln]], axis=1).reshape(-1).astype(np.complex64)
    for nn in (8192,):
        x = crandn(rng, nn)
        g["fftx%d" % nn] = x
        g["fft_fwd%d" % nn] = np.fft.fft(x.astype(np.complex128)).astype(np.complex64)
    np.savez_compressed(os.path.join(HERE, "widen_golden.npz"), **g)


def independent_golden():
    """Fixtures for the two blocks the reference holds nothing for (clPolyphaseChannelizer, clXEngine), produced by implementations that
    are NOT this repository's: scipy.signal.upfirdn (mix down, low-pass FIR, decimate -- the textbook channelizer, one channel at a
    time), numpy.einsum on integers / complex128 with numpy.tril_indices for the baseline order, and scipy.signal.correlate's zero lag
    for single baselines.  The oracle and the GPU path are then checked against data no author of either wrote the arithmetic for."""
    import scipy.signal as sig
    rng = np.random.default_rng(20260929)
    g = {}

    def pfb_scipy(taps, M, R, buf_items, chmap, xh):
        """u_i[c] = sum_k h[k] x[n_i - k] exp(j 2 pi c (k + i (M - R)) / M), n_i = i R + K - 1, as
        exp(j 2 pi c n_i / M) * (h * (x exp(-j 2 pi c n / M)))[n_i] * exp(j 2 pi c i (M - R) / M): mix, FIR, decimate by R."""
        K = taps.size
        nsteps = buf_items // R
        n = np.arange(xh.size)
        pad = (-(K - 1)) % R          # zeros in front so that the wanted samples fall on the decimation grid
        j0 = (K - 1 + pad) // R
        out = np.zeros((nsteps, len(chmap)), np.complex128)
        i = np.arange(nsteps)
        ni = i * R + K - 1
        for q, c in enumerate(chmap):
            xm = xh.astype(np.complex128) * np.exp(-2j * np.pi * c * n / M)
            y = sig.upfirdn(taps.astype(np.float64), np.concatenate([np.zeros(pad, np.complex128), xm]), up=1, down=R)
            out[:, q] = y[j0:j0 + nsteps] * np.exp(2j * np.pi * c * ni / M) * np.exp(2j * np.pi * c * i * (M - R) / M)
        return out.reshape(-1).astype(np.complex64)

    def pfb_closed(taps, M, R, buf_items, chmap, xh):  # (the definition, only to make sure the scipy form above is the same quantity)
        K = taps.size
        kk = np.arange(K)
        out = np.zeros((buf_items // R, len(chmap)), np.complex128)
        for i in range(buf_items // R):
            seg = xh[i * R + K - 1 - kk].astype(np.complex128) * taps.astype(np.float64)
            for q, c in enumerate(chmap):
                out[i, q] = np.sum(seg * np.exp(2j * np.pi * c * (kk + i * (M - R)) / M))
        return out.reshape(-1).astype(np.complex64)

    cases = {
        "pa": (firdes_low_pass64(1.0, 300e3, 48e3, 5e3).astype(np.float32), 3, 2, 60, [0, 1, 2]),                    # the reference flowgraph's case
        "pb": (np.concatenate([firdes_low_pass64(1.0, 64.0, 0.5, 0.0753), [0.0]]).astype(np.float32), 64, 64, 64 * 8, list(range(64))),  # BASELINE config 4
        "pc": (rng.standard_normal(8 * 6).astype(np.float32), 8, 4, 96, [5, 0, 7, 2, 2, 1]),                           # 2-fold oversampled, permuted partial map
        "pd": (rng.standard_normal(16 * 4 + 5).astype(np.float32), 16, 4, 64, list(range(16))),                         # 4-fold oversampled, ragged tap count
    }
    for key, (taps, M, R, buf, cm) in cases.items():
        xh = crandn(rng, buf - R + taps.size)
        y = pfb_scipy(taps, M, R, buf, cm, xh)
        ref = pfb_closed(taps, M, R, buf, cm, xh)
        assert np.abs(y - ref).max() <= 2e-6 * np.abs(ref).max(), key
        g[key + "_taps"], g[key + "_x"], g[key + "_cfg"], g[key + "_chmap"], g[key + "_y"] = taps, xh, np.array([M, R, buf], np.int32), np.array(cm, np.int32), y

    def xeng_einsum(x, N, F, npol, T):
        """x: int8 [T][N][F][npol][2] -> exact integer sums [F][B][npol^2] (re, im) by numpy.einsum on int64; baselines by numpy.tril_indices."""
        rows = x.reshape(T, N, F, npol, 2).transpose(0, 1, 3, 2, 4).reshape(T, N * npol, F, 2).astype(np.int64)  # [t][row = s * npol + p][f]
        I, Q = rows[..., 0], rows[..., 1]
        re = np.einsum("trf,tuf->fru", I, I) + np.einsum("trf,tuf->fru", Q, Q)      # z_r conj(z_u): re = I I + Q Q
        im = np.einsum("trf,tuf->fru", Q, I) - np.einsum("trf,tuf->fru", I, Q)      #                im = Q I - I Q
        s1, s2 = np.tril_indices(N)                                                    # k = s1 (s1 + 1) / 2 + s2, s1 >= s2
        out_re = np.zeros((F, s1.size, npol * npol), np.int64)
        out_im = np.zeros_like(out_re)
        for p1 in range(npol):
            for p2 in range(npol):
                out_re[:, :, p1 * npol + p2] = re[:, s1 * npol + p1, s2 * npol + p2]
                out_im[:, :, p1 * npol + p2] = im[:, s1 * npol + p1, s2 * npol + p2]
        return out_re, out_im

    for key, (N, F, T, npol) in {"xa": (6, 5, 40, 1), "xb": (5, 4, 24, 2), "xc": (9, 3, 33, 1)}.items():
        x = rng.integers(-128, 128, size=(T, N, F, npol, 2), dtype=np.int64).astype(np.int8)
        sre, sim = xeng_einsum(x, N, F, npol, T)
        g[key + "_x"], g[key + "_cfg"] = x.reshape(-1), np.array([N, F, T, npol], np.int32)
        g[key + "_sum_re"], g[key + "_sum_im"] = sre.reshape(-1), sim.reshape(-1)
        # the same through complex128 on the scaled samples (the reference kernel's arithmetic order: scale first, lib/clXEngine_impl.cc:859-867)
        z = (x[..., 0].astype(np.float64) + 1j * x[..., 1].astype(np.float64)) / 127.0
        rows = z.transpose(0, 1, 3, 2).reshape(T, N * npol, F)
        v = np.einsum("trf,tuf->fru", rows, rows.conj())
        s1, s2 = np.tril_indices(N)
        vv = np.zeros((F, s1.size, npol * npol), np.complex128)
        for p1 in range(npol):
            for p2 in range(npol):
                vv[:, :, p1 * npol + p2] = v[:, s1 * npol + p1, s2 * npol + p2]
        g[key + "_y"] = vv.reshape(-1).astype(np.complex64)
        assert np.abs(vv.reshape(-1) - (sre.reshape(-1) + 1j * sim.reshape(-1)) / 127.0 ** 2).max() < 1e-9
    # single baselines as the zero lag of scipy.signal.correlate (which conjugates its second argument)
    N, F, T, npol = (int(v) for v in g["xa_cfg"])
    x = g["xa_x"].reshape(T, N, F, 2).astype(np.float64)
    z = (x[..., 0] + 1j * x[..., 1]) / 127.0
    picks = [(3, 1, 2), (5, 5, 0), (4, 0, 4)]
    g["xa_picks"] = np.array(picks, np.int32)
    g["xa_pick_vals"] = np.array([sig.correlate(z[:, s1, f], z[:, s2, f], mode="valid")[0] for s1, s2, f in picks]).astype(np.complex64)
    # ---- clFFT and the filters: library implementations as well (scipy.fft = pocketfft, scipy.signal.windows, scipy.signal.firwin /
    # upfirdn / lfilter), so that no block's general-input parity rests on arithmetic written for this repository
    import scipy.fft as sfft
    import scipy.signal.windows as swin
    for key, n, frames in (("ta", 4096, 3), ("tb", 1000, 2), ("tc", 4099, 1), ("td", 64, 5)):     # BASELINE config 2's length; 2^3 5^3; a prime (chirp-z); small
        x = crandn(rng, n * frames)
        w = swin.blackman(n, sym=True).astype(np.float32)
        X = sfft.fft(x.astype(np.complex128).reshape(frames, n) * w.astype(np.float64), axis=1)
        g[key + "_x"], g[key + "_win"] = x, w
        g[key + "_fwd_win_shift"] = sfft.fftshift(X, axes=1).reshape(-1).astype(np.complex64)
        g[key + "_fwd"] = sfft.fft(x.astype(np.complex128).reshape(frames, n), axis=1).reshape(-1).astype(np.complex64)
        g[key + "_inv"] = (sfft.ifft(x.astype(np.complex128).reshape(frames, n), axis=1) * n).reshape(-1).astype(np.complex64)  # unscaled inverse, as clFFT
    taps = sig.firwin(65, 0.2, window="hamming").astype(np.float32)                                   # 65 taps: BASELINE config 3's length
    ctaps = (taps * np.exp(2j * np.pi * 0.11 * np.arange(65))).astype(np.complex64)
    long_taps = sig.firwin(3001, 0.05, window=("kaiser", 7.0)).astype(np.float32)                     # past 2048: the partitioned fast convolution
    x = crandn(rng, 6000)
    g["fa_x"], g["fa_taps"], g["fa_ctaps"], g["fa_long_taps"] = x, taps, ctaps, long_taps
    for d in (1, 2, 3, 10):
        g["fa_y_d%d" % d] = sig.upfirdn(taps.astype(np.float64), x.astype(np.complex128), up=1, down=d)[:x.size // d].astype(np.complex64)
        g["fa_yc_d%d" % d] = sig.upfirdn(ctaps.astype(np.complex128), x.astype(np.complex128), up=1, down=d)[:x.size // d].astype(np.complex64)
    g["fa_y_lfilter"] = sig.lfilter(taps.astype(np.float64), [1.0], x.astype(np.complex128)).astype(np.complex64)       # the same numbers by another routine
    assert np.abs(g["fa_y_lfilter"] - g["fa_y_d1"]).max() <= 1e-6 * np.abs(g["fa_y_d1"]).max()
    g["fa_y_long"] = sig.fftconvolve(x.astype(np.complex128), long_taps.astype(np.float64))[:x.size].astype(np.complex64)
    # ---- clxcorrelate_fft_vcf, time-series inputs: circular cross-correlation as scipy.signal.correlate's direct-form linear correlation folded
    # onto N lags (lag k and lag k - N land on the same output), magnitude, half swap
    n, nfr = 256, 2
    xin = [crandn(rng, n * nfr) for _ in range(3)]
    for s_ in (1, 2):
        rows = []
        for fr in range(nfr):
            a, b = (v[fr * n:(fr + 1) * n].astype(np.complex128) for v in (xin[0], xin[s_]))
            full = sig.correlate(a, b, mode="full", method="direct")          # full[k + n - 1] = sum_m a[m + k] conj(b[m]), k = -(n-1) .. n-1
            circ = full[n - 1:].copy()
            circ[1:] += full[:n - 1]
            rows.append(sfft.fftshift(np.abs(circ * n)))                      # the block's transform pair is unscaled: n x the correlation
        g["xc_out%d" % s_] = np.concatenate(rows).astype(np.float32)
    for i, v in enumerate(xin):
        g["xc_in%d" % i] = v
    np.savez_compressed(os.path.join(HERE, "independent_golden.npz"), **g)


def cli_golden():
    """Inputs the reference's own timing CLIs build (closed forms), with float64 expectations:
    * lib/test-clfilter.cc:98-100 + :76-80: taps i/1000 over a constant (1.0, 0.5) stream -> every output is
      (1 + 0.5j) * sum(taps) = (1 + 0.5j) * ntaps (ntaps - 1) / 2000  (a true closed form, kept in cli_kat.json);
    * lib/test_clenabled.cc:835-851: x[i] = (sin(w i), cos(w i)), w = 2 pi / N in FLOAT arithmetic, through the Blackman
      window + fftshift of BASELINE config 1 (expected spectrum: float64 numpy of the float32 inputs)."""
    g = {}
    n = 4096
    w = np.float32(2 * np.pi * 10) / np.float32(n * 10)            # float frequency_sampling = fftDataSize * frequency_signal
    ph = (np.float32(0.0) + w * np.arange(n, dtype=np.float32)).astype(np.float32)
    x = (np.sin(ph).astype(np.float32) + 1j * np.cos(ph).astype(np.float32)).astype(np.complex64)
    k = np.arange(n)
    win = (0.42 - 0.5 * np.cos(2 * np.pi * k / (n - 1)) + 0.08 * np.cos(4 * np.pi * k / (n - 1))).astype(np.float32)
    g["tone4096_x"] = x
    g["tone4096_win"] = win
    X = np.fft.fft(x.astype(np.complex128) * win.astype(np.float64))
    g["tone4096_fwd_win_shift"] = np.fft.fftshift(X).astype(np.complex64)
    g["tone4096_fwd"] = np.fft.fft(x.astype(np.complex128)).astype(np.complex64)
    np.savez_compressed(os.path.join(HERE, "cli_golden.npz"), **g)
    kat = {"_source": "closed forms of the inputs the reference's timing CLIs build",
           "filter_ramp_taps": [{"ntaps": nt, "tap_i": "i/1000", "input": [1.0, 0.5],
                                 "expect": [nt * (nt - 1) / 2000.0, 0.5 * nt * (nt - 1) / 2000.0],
                                 "ref": "lib/test-clfilter.cc:76-80,98-100"} for nt in (3, 65, 128, 1000)],
           "fft_tone_4096": {"peak_bin_unshifted": n - 1, "peak": [0.0, float(n)], "ref": "lib/test_clenabled.cc:835-851"}}
    with open(os.path.join(HERE, "cli_kat.json"), "w") as f:
        json.dump(kat, f, indent=1)


def main():
    kat = {
        "_source": "reference known-answer tests and SURVEY.md section 8(c)",
        "mathop_multiply": {"a": [1.0, 0.5], "b": [1.0, 0.5], "n": 8192, "expect": [0.75, 1.0],
                            "ref": "lib/test_clenabled.cc:1596-1600"},
        "mathconst_multiply": {"a": [1.0, 0.5], "k": 2.0, "expect": [2.0, 1.0], "ref": "lib/test_clenabled.cc:1351-1356"},
        "fft_tone": {"n": 2048, "peak_bin": 2047, "peak": [0.0, 2048.0], "others_abs_max": 1e-3,
                     "ref": "lib/clFFT_impl.cc:361-455 (FFTValidationTest input :369-377)"},
        "firdes_low_pass_65": {"args": [1.0, 10e6, 1e6, 372000.0], "ntaps": 65, "t0": 0.000756795635, "t32": 0.199991778,
                               "ref": "lib/firdes.cc:92-137 compiled by the survey"},
        "firdes_low_pass_2047": {"args": [1.0, 64.0, 0.5, 0.0753], "ntaps": 2047, "t0": -1.22261542e-06,
                                 "t1023": 0.0156404767, "ref": "lib/firdes.cc:92-137 compiled by the survey"},
        "firdes_low_pass_145": {"args": [1.0, 300e3, 48e3, 5e3], "ntaps": 145,
                                "ref": "examples/test_flowgraphs/OpenCL_Test-PolyphaseChannelizer.grc:47-75"},
        "window_blackman_4096": {"i1": 2.01165676e-07, "i2048": 0.999999762, "ref": "lib/window.cc:166-170"},
        "fft_filter_sizes_65": {"fftsize": 256, "nsamples": 192, "ref": "lib/fft_filter.cc:72-97"},
    }
    with open(os.path.join(HERE, "kat.json"), "w") as f:
        json.dump(kat, f, indent=1)

    rng = np.random.default_rng(20260928)

    # ---- FFT: seeded frames vs float64 numpy.fft -------------------------------
    fft = {}
    for n in (8, 64, 1024, 4096):
        x = crandn(rng, 2 * n)
        X = np.fft.fft(x.astype(np.complex128).reshape(2, n), axis=1)
        fft["x%d" % n] = x
        fft["fwd%d" % n] = X.reshape(-1).astype(np.complex64)
        fft["inv%d" % n] = (np.fft.ifft(x.astype(np.complex128).reshape(2, n), axis=1) * n).reshape(-1).astype(np.complex64)
    n = 4096
    k = np.arange(n)
    win = (0.42 - 0.5 * np.cos(2 * np.pi * k / (n - 1)) + 0.08 * np.cos(4 * np.pi * k / (n - 1)))
    fft["blackman4096"] = win.astype(np.float32)
    xw = fft["x4096"].astype(np.complex128).reshape(2, n) * win.astype(np.float32).astype(np.float64)
    fft["fwd_win_shift4096"] = np.fft.fftshift(np.fft.fft(xw, axis=1), axes=1).reshape(-1).astype(np.complex64)
    # reverse + shift (no window): halves swapped on input, unnormalised inverse
    xs = np.fft.ifftshift(fft["x4096"].astype(np.complex128).reshape(2, n), axes=1)
    fft["inv_shift4096"] = (np.fft.ifft(xs, axis=1) * n).reshape(-1).astype(np.complex64)
    # the reference's one-cycle tone (lib/clFFT_impl.cc:369-377, lib/test_clenabled.cc:835-851)
    t = np.arange(2048)
    fft["tone2048"] = (np.sin(2 * np.pi * t / 2048) + 1j * np.cos(2 * np.pi * t / 2048)).astype(np.complex64)
    # real input
    xr = rng.standard_normal(2 * 1024).astype(np.float32)
    fft["xr1024"] = xr
    fft["fwd_real1024"] = np.fft.fft(xr.astype(np.float64).reshape(2, 1024), axis=1).reshape(-1).astype(np.complex64)
    np.savez_compressed(os.path.join(HERE, "fft_golden.npz"), **fft)

    # ---- filters: float64 convolution -----------------------------------------------
    flt = {}
    taps = firdes_low_pass64(1.0, 10e6, 1e6, 372000.0).astype(np.float32)
    assert taps.size == 65
    x = crandn(rng, 8 * 192)
    y = np.convolve(x.astype(np.complex128), taps.astype(np.float64))[: x.size]
    flt["taps65"], flt["x"], flt["y_d1"] = taps, x, y.astype(np.complex64)
    flt["y_d2"], flt["y_d3"] = y[::2].astype(np.complex64), y[::3].astype(np.complex64)
    ctaps = (taps.astype(np.float64) * np.exp(1j * np.pi * np.arange(65) / 8)).astype(np.complex64)
    yc = np.convolve(x.astype(np.complex128), ctaps.astype(np.complex128))[: x.size]
    flt["ctaps65"], flt["yc_d1"], flt["yc_d2"] = ctaps, yc.astype(np.complex64), yc[::2].astype(np.complex64)
    t7 = np.array([0.001 * i for i in range(1, 8)], np.float32)  # taps i/1000, lib/test-clfilter.cc:98-100
    flt["taps7"], flt["y7_d1"] = t7, np.convolve(x.astype(np.complex128), t7.astype(np.float64))[: x.size].astype(np.complex64)
    np.savez_compressed(os.path.join(HERE, "filter_golden.npz"), **flt)

    # ---- polyphase channelizer: closed form of SURVEY App. A.4 in float64 ---------------
    pfb = {}

    def pfb_closed(taps, M, R, buf_items, chmap, xh):
        K = taps.size
        nsteps = buf_items // R
        out = np.zeros((nsteps, len(chmap)), np.complex128)
        kk = np.arange(K)
        for i in range(nsteps):
            seg = xh[i * R + K - 1 - kk].astype(np.complex128) * taps.astype(np.float64)
            for q, c in enumerate(chmap):
                out[i, q] = np.sum(seg * np.exp(2j * np.pi * c * (kk + i * (M - R)) / M))
        return out.reshape(-1).astype(np.complex64)

    # (a) the reference flowgraph case: M=3, oversampled R=2, 145 taps
    t145 = firdes_low_pass64(1.0, 300e3, 48e3, 5e3).astype(np.float32)
    assert t145.size == 145
    M, R, buf = 3, 2, 60
    xh = crandn(rng, buf - R + t145.size)
    pfb["a_taps"], pfb["a_x"], pfb["a_cfg"] = t145, xh, np.array([M, R, buf], np.int32)
    pfb["a_chmap"] = np.array([0, 1, 2], np.int32)
    pfb["a_y"] = pfb_closed(t145, M, R, buf, [0, 1, 2], xh)
    # (b) BASELINE config 4 shape: M=64, R=64, 2048 taps (2047 + one zero), partial channel map
    t2048 = np.concatenate([firdes_low_pass64(1.0, 64.0, 0.5, 0.0753), [0.0]]).astype(np.float32)
    assert t2048.size == 2048
    M, R, buf = 64, 64, 64 * 6
    xh = crandn(rng, buf - R + 2048)
    pfb["b_taps"], pfb["b_x"], pfb["b_cfg"] = t2048, xh, np.array([M, R, buf], np.int32)
    pfb["b_chmap"] = np.arange(64, dtype=np.int32)
    pfb["b_y"] = pfb_closed(t2048, M, R, buf, list(range(64)), xh)
    # (c) oversampled power-of-two case with a permuted, partial map: M=8, R=4
    t8 = rng.standard_normal(8 * 5 + 3).astype(np.float32)
    M, R, buf = 8, 4, 64
    xh = crandn(rng, buf - R + t8.size)
    cm = [7, 0, 3, 3, 5]
    pfb["c_taps"], pfb["c_x"], pfb["c_cfg"] = t8, xh, np.array([M, R, buf], np.int32)
    pfb["c_chmap"] = np.array(cm, np.int32)
    pfb["c_y"] = pfb_closed(t8, M, R, buf, cm, xh)
    np.savez_compressed(os.path.join(HERE, "pfb_golden.npz"), **pfb)

    # ---- X-engine: exact integer sums ---------------------------------------------------
    xe = {}

    def xeng_exact(x, N, F, npol, T):
        """x: int array [T][N][F][npol][2] -> float64 complex [F][B][npol*npol] (unscaled integer sums)."""
        z = x[..., 0].astype(np.float64) + 1j * x[..., 1].astype(np.float64)
        B = N * (N + 1) // 2
        out = np.zeros((F, B, npol * npol), np.complex128)
        for s1 in range(N):
            for s2 in range(s1 + 1):
                k = s1 * (s1 + 1) // 2 + s2
                for p1 in range(npol):
                    for p2 in range(npol):
                        out[:, k, p1 * npol + p2] = np.sum(z[:, s1, :, p1] * np.conj(z[:, s2, :, p2]), axis=0)
        return out

    N, F, T = 4, 8, 16
    for npol in (1, 2):
        x = rng.integers(-127, 128, size=(T, N, F, npol, 2), dtype=np.int64).astype(np.int8)
        s = xeng_exact(x, N, F, npol, T)
        xe["i8_p%d_x" % npol] = x.reshape(-1)
        xe["i8_p%d_sum_re" % npol] = s.real.reshape(-1).astype(np.int64)
        xe["i8_p%d_sum_im" % npol] = s.imag.reshape(-1).astype(np.int64)
    xe["cfg"] = np.array([N, F, T], np.int32)
    # complex float path: same definition in float64
    xc = crandn(rng, T * N * F * 2).reshape(T, N, F, 2)
    z = xc.astype(np.complex128)
    B = N * (N + 1) // 2
    o = np.zeros((F, B, 4), np.complex128)
    for s1 in range(N):
        for s2 in range(s1 + 1):
            k = s1 * (s1 + 1) // 2 + s2
            for p1 in range(2):
                for p2 in range(2):
                    o[:, k, p1 * 2 + p2] = np.sum(z[:, s1, :, p1] * np.conj(z[:, s2, :, p2]), axis=0)
    xe["cf_p2_x"], xe["cf_p2_y"] = xc.reshape(-1), o.reshape(-1).astype(np.complex64)
    # packed 4-bit: LUT incl. code 8 -> 0 (lib/clXEngine_impl.cc:833), scale 1/7
    lut = np.array([0, 1, 2, 3, 4, 5, 6, 7, 0, -7, -6, -5, -4, -3, -2, -1], np.float64)
    pk = rng.integers(0, 256, size=(T, N, F, 2), dtype=np.int64).astype(np.uint8)
    pk[0, 0, 0, 0] = 0x88  # force the code-8 case
    zz = (lut[pk >> 4] + 1j * lut[pk & 15]) / 7.0  # [T][N][F][pol]
    o = np.zeros((F, B, 4), np.complex128)
    for s1 in range(N):
        for s2 in range(s1 + 1):
            k = s1 * (s1 + 1) // 2 + s2
            for p1 in range(2):
                for p2 in range(2):
                    o[:, k, p1 * 2 + p2] = np.sum(zz[:, s1, :, p1] * np.conj(zz[:, s2, :, p2]), axis=0)
    xe["p4_x"], xe["p4_y"] = pk.reshape(-1), o.reshape(-1).astype(np.complex64)
    np.savez_compressed(os.path.join(HERE, "xengine_golden.npz"), **xe)
    widen_golden()
    cli_golden()
    independent_golden()

    tot = sum(os.path.getsize(os.path.join(HERE, f)) for f in os.listdir(HERE) if f.endswith((".npz", ".json")))
    print("golden fixtures written, %.1f KiB" % (tot / 1024))


if __name__ == "__main__":
    main()
