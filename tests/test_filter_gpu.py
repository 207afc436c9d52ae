"""GPU parity: clFilter / clComplexFilter (fast-convolution and direct-form kernels)
through the C ABI vs the oracle's fir_filter / fft_filter restatements and the golden vectors.
Semantics under test: y[m] = sum_k h[k] x[m*decim - k] on GNU Radio's history-prefixed buffer."""
import numpy as np
import pytest

from conftest import GPU_ARGS, crandn, golden, relerr

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _hist(x, ntaps):
    return np.concatenate([np.zeros(ntaps - 1, np.complex64), x])


@pytest.mark.parametrize("use_time", [False, True])
def test_golden_65_taps_all_decimations(gpu, use_time):
    g = golden("filter_golden.npz")
    x, t = g["x"], g["taps65"]
    xh = _hist(x, 65)
    for d in (1, 2, 3):
        blk = gpu.clFilter(*GPU_ARGS, d, t, 1, 0, use_time)
        n = x.size // d
        y = np.empty(n, np.complex64)
        assert blk.work(n, [xh], [y]) == n
        assert relerr(y, g["y_d%d" % d][:n]) <= TOL
    blk = gpu.clFilter(*GPU_ARGS, 1, g["taps7"], 1, 0, use_time)
    y = np.empty(x.size, np.complex64)
    blk.work(x.size, [_hist(x, 7)], [y])
    assert relerr(y, g["y7_d1"]) <= TOL


@pytest.mark.parametrize("use_time", [False, True])
def test_golden_complex_taps(gpu, use_time):
    g = golden("filter_golden.npz")
    x, t = g["x"], g["ctaps65"]
    xh = _hist(x, 65)
    for d in (1, 2):
        blk = gpu.clComplexFilter(*GPU_ARGS, d, t, 1, 0, use_time=use_time)
        n = x.size // d
        y = np.empty(n, np.complex64)
        blk.work(n, [xh], [y])
        assert relerr(y, g["yc_d%d" % d][:n]) <= TOL


@pytest.mark.parametrize("use_time", [False, True])
@pytest.mark.parametrize("ntaps", [1, 2, 3, 8, 9, 17, 30, 64, 65, 100, 128, 129, 300, 600, 1000, 2048, 2049, 3000, 4096, 4097, 6001, 9000])  # > 2048: partitioned into <= 2048-tap segments in FFT mode
def test_vs_oracle_fir_various_lengths(gpu, oracle, ntaps, use_time):
    rng = np.random.default_rng(ntaps)
    taps = (rng.standard_normal(ntaps) / np.sqrt(ntaps)).astype(np.float32)
    n = 5000 + ntaps  # ragged: not a multiple of any block length
    xh = crandn(rng, n + ntaps - 1)  # non-zero history
    blk = gpu.clFilter(*GPU_ARGS, 1, taps, 1, 0, use_time)
    assert blk.history() == ntaps
    y = np.empty(n, np.complex64)
    blk.work(n, [xh], [y])
    assert relerr(y, oracle.fir_ccf(taps, xh, n)) <= TOL


@pytest.mark.parametrize("decim", [1, 3])
def test_long_filter_partitioned_streaming(gpu, oracle, decim):
    """5000 taps = three segments of the NF = 4096 kernel accumulating into y; two consecutive calls with history equal one
    call and the oracle, with and without decimation, through the device-resident path as well."""
    import torch
    rng = np.random.default_rng(50 + decim)
    ntaps = 5000
    taps = (rng.standard_normal(ntaps) / np.sqrt(ntaps)).astype(np.float32)
    n = 7000
    x = crandn(rng, 2 * n * decim + ntaps - 1)
    blk = gpu.clFilter(*GPU_ARGS, decim, taps)
    assert blk.fftsize() == 4096
    ya, yb, yw = (np.empty(m, np.complex64) for m in (n, n, 2 * n))
    blk.work(n, [x[:n * decim + ntaps - 1]], [ya])
    blk.work(n, [x[n * decim:]], [yb])
    blk.work(2 * n, [x], [yw])
    assert relerr(np.concatenate([ya, yb]), yw) <= 2e-6  # same samples, block boundaries at different places
    ref = oracle.fir_ccf(taps, x, 2 * n * decim)[::decim][:2 * n]
    assert relerr(yw, ref) <= TOL
    xd = torch.from_numpy(x.view(np.float32).reshape(-1, 2)).cuda()
    yd = torch.empty(2 * n, 2, device="cuda")
    blk.work_device(2 * n, [xd], [yd])
    assert relerr(yd.cpu().numpy().view(np.complex64).reshape(-1), ref) <= TOL


@pytest.mark.parametrize("wgs", ["2", "3", "1000"])
@pytest.mark.parametrize("ntaps,decim", [(2049, 1), (3000, 1), (4096, 2), (4097, 1), (6001, 3), (8000, 1), (8192, 1), (9000, 2), (10240, 1), (10241, 1)])
def test_long_filter_runs_of_blocks(gpu, oracle, monkeypatch, ntaps, decim, wgs):
    """2049 .. 10240 taps run the uniformly partitioned kernel (segments of 2048 taps = the block length; a workgroup walks a
    run of blocks and carries the spectra of the previous input blocks in registers).  A small grid (MI355_OLS_UPS_WGS) makes
    the runs many blocks long at a test-sized call; 1000 workgroups = one block each (every block warms up its own ring).
    10241 taps = six segments: the general partitioned kernel."""
    monkeypatch.setenv("MI355_OLS_UPS_WGS", wgs)
    rng = np.random.default_rng(ntaps + decim)
    taps = (rng.standard_normal(ntaps) / np.sqrt(ntaps)).astype(np.float32)
    n = 2048 * 13 // decim + 77  # 13-14 blocks, ragged end
    xh = crandn(rng, n * decim + ntaps - 1)
    blk = gpu.clFilter(*GPU_ARGS, decim, taps)
    assert blk.fftsize() == 4096
    y = np.empty(n, np.complex64)
    assert blk.work(n, [xh], [y]) == n
    ref = oracle.fir_ccf(taps, xh, n, decim)
    assert relerr(y, ref) <= TOL
    monkeypatch.setenv("MI355_OLS_UPS", "0")  # the general partitioned kernel on the same call
    y2 = np.empty(n, np.complex64)
    assert blk.work(n, [xh], [y2]) == n
    assert relerr(y2, ref) <= TOL


def test_long_complex_filter_partitioned(gpu, oracle):
    """clComplexFilter, 4500 complex taps in the fast-convolution mode = three accumulating segments; tiny and ragged calls."""
    rng = np.random.default_rng(77)
    ntaps = 4500
    taps = ((rng.standard_normal(ntaps) + 1j * rng.standard_normal(ntaps)) / np.sqrt(2 * ntaps)).astype(np.complex64)
    blk = gpu.clComplexFilter(*GPU_ARGS, 1, taps, 1, 0, use_time=False)
    assert blk.fftsize() == 4096
    for n in (1, 17, 2049, 6000):
        x = crandn(rng, n + ntaps - 1)
        y = np.empty(n, np.complex64)
        assert blk.work(n, [x], [y]) == n
        assert relerr(y, oracle.fir_ccc(taps, x, n)) <= TOL, n


def test_reference_fft_sizes_and_stateful_oracle(gpu, oracle):
    """The fused overlap-save kernel equals the reference's stateful overlap-add
    (lib/fft_filter.cc:133-175) run over the same stream from a zero tail."""
    taps = oracle.firdes_low_pass(1.0, 10e6, 1e6, 372000.0)
    blk = gpu.clFilter(*GPU_ARGS, 1, taps)
    f = oracle.FFTFilter(1, taps)
    assert (blk.fftsize(), f.fftsize, f.nsamples) == (256, 256, 192)  # lib/fft_filter.cc:72-97
    rng = np.random.default_rng(5)
    x = crandn(rng, 192 * 171)  # the reference's block count for a 32768-sample buffer
    y = np.empty_like(x)
    blk.work(x.size, [_hist(x, taps.size)], [y])
    assert relerr(y, f.filter(x.size, x)) <= TOL


def test_streaming_calls_equal_one_call(gpu, oracle):
    """work() is stateless given history: consecutive 32768-sample calls (BASELINE config 3)
    reproduce one long call."""
    taps = oracle.firdes_low_pass(1.0, 10e6, 1e6, 372000.0)
    rng = np.random.default_rng(6)
    x = crandn(rng, 4 * 32768)
    xh = _hist(x, 65)
    for use_time in (False, True):
        blk = gpu.clFilter(*GPU_ARGS, 1, taps, 1, 0, use_time)
        whole = np.empty_like(x)
        blk.work(x.size, [xh], [whole])
        parts = np.empty_like(x)
        for c in range(4):
            blk.work(32768, [xh[c * 32768:(c + 1) * 32768 + 64]], [parts[c * 32768:(c + 1) * 32768]])
        assert relerr(parts, whole) <= 2e-6
        assert relerr(whole, oracle.fir_ccf(taps, xh, x.size)) <= TOL


def test_set_taps_and_taps_roundtrip(gpu, oracle):
    rng = np.random.default_rng(7)
    t1 = rng.standard_normal(33).astype(np.float32)
    t2 = rng.standard_normal(200).astype(np.float32)
    xh = crandn(rng, 3000 + 199)
    for use_time in (False, True):
        blk = gpu.clFilter(*GPU_ARGS, 1, t1, 1, 0, use_time)
        assert np.array_equal(blk.taps(), t1)
        blk.set_taps2(t2)
        assert np.array_equal(blk.taps(), t2) and blk.ntaps() == 200
        y = np.empty(3000, np.complex64)
        blk.work(3000, [xh], [y])
        assert relerr(y, oracle.fir_ccf(t2, xh, 3000)) <= TOL


def test_impulse_response_is_the_taps(gpu):
    taps = np.arange(1, 66, dtype=np.float32) / 1000  # lib/test-clfilter.cc:98-100 style taps i/1000
    x = np.zeros(1000, np.complex64)
    x[10] = 1 + 2j
    for use_time in (False, True):
        y = np.empty_like(x)
        gpu.clFilter(*GPU_ARGS, 1, taps, 1, 0, use_time).work(x.size, [_hist(x, 65)], [y])
        exp = np.zeros_like(x)
        exp[10:75] = taps * (1 + 2j)
        assert relerr(y, exp) <= TOL


def test_host_path_multi_chunk_with_decimation(gpu, oracle):
    rng = np.random.default_rng(8)
    taps = oracle.firdes_low_pass(1.0, 10e6, 1e6, 372000.0)
    d, n = 2, (1 << 20) + 777  # > 2 staging chunks
    xh = crandn(rng, n * d + 64)
    y = np.empty(n, np.complex64)
    gpu.clFilter(*GPU_ARGS, d, taps).work(n, [xh], [y])
    for o0 in (0, 524288 - 50, n - 3000):  # around chunk boundaries and the ragged end
        ref = oracle.fir_ccf(taps, xh[o0 * d:], min(3000, n - o0), d)
        assert relerr(y[o0:o0 + ref.size], ref) <= TOL


def test_device_path_full_size_two_kernels_agree(gpu, oracle):
    """BASELINE config 3 at full size: 2^27 samples device resident, 65 taps, decim 1.
    The fast-convolution kernel and the direct-form kernel are independent implementations;
    they must agree everywhere, and a sampled window must match the oracle."""
    import torch
    taps = oracle.firdes_low_pass(1.0, 10e6, 1e6, 372000.0)
    n = 1 << 27
    g = torch.Generator(device="cuda").manual_seed(99)
    x = torch.randn(n + 64, 2, device="cuda", generator=g)
    yf = torch.empty(n, 2, device="cuda")
    yt = torch.empty(n, 2, device="cuda")
    gpu.clFilter(*GPU_ARGS, 1, taps, 1, 0, False).work_device(n, [x], [yf])
    gpu.clFilter(*GPU_ARGS, 1, taps, 1, 0, True).work_device(n, [x], [yt])
    torch.cuda.synchronize()
    scale = yt.abs().max().item()
    assert (yf - yt).abs().max().item() <= 4e-6 * scale
    for o0 in (0, 12345678, n - 5000):
        xs = x[o0:o0 + 5000 + 64].cpu().numpy().view(np.complex64).reshape(-1)
        ys = yf[o0:o0 + 5000].cpu().numpy().view(np.complex64).reshape(-1)
        assert relerr(ys, oracle.fir_ccf(taps, xs, 5000)) <= TOL


@pytest.mark.parametrize("use_time", [False, True])
@pytest.mark.parametrize("decim", [2, 4, 7, 8, 9, 16, 20, 50, 100, 700])
def test_decimations_both_modes(gpu, oracle, decim, use_time):
    """decimation <= 8 runs in the tiled direct-form kernel, larger ones from a span staged in LDS (k_fir_dec_lds), a decimation above
    eight filter lengths (700) one output per thread; the FFT mode decimates on the store.  y[m] = (h * x)[m * decim]."""
    rng = np.random.default_rng(decim)
    ntaps = 77
    taps = (rng.standard_normal(ntaps) / np.sqrt(ntaps)).astype(np.float32)
    nout = 3001
    xh = crandn(rng, nout * decim + ntaps - 1)
    blk = gpu.clFilter(*GPU_ARGS, decim, taps, 1, 0, use_time)
    y = np.empty(nout, np.complex64)
    assert blk.work(nout, [xh], [y]) == nout
    full = oracle.fir_ccf(taps, xh, nout * decim)
    assert relerr(y, full[::decim][:nout]) <= TOL


@pytest.mark.parametrize("ntaps,decim,ctaps", [(33, 12, True), (65, 40, False), (200, 100, False), (1000, 12, False), (1500, 25, True), (4000, 10, False), (5000, 9, False)])
def test_large_decimations_time_domain_shapes(gpu, oracle, monkeypatch, ntaps, decim, ctaps):
    """The LDS-staged decimating kernel: complex taps, filters that fill most of a tile's span (4000 of 8192 samples), one that does not
    fit (5000 taps: the per-output kernel), a ragged last tile -- and each of the three kernels for decimations above 8 (LDS-staged, per
    output, every undecimated output on the matrix cores) forced on the same inputs (MI355_FIR_DEC_KERNEL)."""
    rng = np.random.default_rng(ntaps + decim)
    nout = 2777
    xh = crandn(rng, nout * decim + ntaps - 1)
    if ctaps:
        taps = (crandn(rng, ntaps) / np.sqrt(ntaps)).astype(np.complex64)
        ref = oracle.fir_ccc(taps, xh, nout * decim)[::decim][:nout]
    else:
        taps = (rng.standard_normal(ntaps) / np.sqrt(ntaps)).astype(np.float32)
        ref = oracle.fir_ccf(taps, xh, nout * decim)[::decim][:nout]
    for force in (None, "lds", "per_output", "all"):  # the block's own choice, then each of the three kernels where it applies (read per call)
        if force:
            monkeypatch.setenv("MI355_FIR_DEC_KERNEL", force)
        blk = gpu.clComplexFilter(*GPU_ARGS, decim, taps, 1, 0, use_time=True) if ctaps else gpu.clFilter(*GPU_ARGS, decim, taps, 1, 0, True)
        y = np.empty(nout, np.complex64)
        assert blk.work(nout, [xh], [y]) == nout
        assert relerr(y, ref) <= TOL, (ntaps, decim, ctaps, force)


def test_zero_outputs_and_bad_args(gpu):
    blk = gpu.clFilter(*GPU_ARGS, 1, [1.0, 2.0])
    e = np.empty(0, np.complex64)
    assert blk.work(0, [np.zeros(1, np.complex64)], [e]) == 0
    with pytest.raises(ValueError):
        blk.work(10, [np.zeros(5, np.complex64)], [np.empty(10, np.complex64)])  # not enough input for the history
    with pytest.raises(gpu.Mi355Error):
        gpu.clFilter(*GPU_ARGS, 0, [1.0])  # decimation < 1


@pytest.mark.parametrize("use_time", [False, True])
def test_reference_cli_ramp_taps_closed_form(gpu, use_time):
    """The reference's filter timing CLI (lib/test-clfilter.cc:76-80,98-100): taps i/1000 over a constant (1.0, 0.5) stream.
    Every output is (1 + 0.5j) * ntaps (ntaps - 1) / 2000 -- a closed form held in tests/golden/cli_kat.json."""
    import json
    import os
    from conftest import GOLDEN
    with open(os.path.join(GOLDEN, "cli_kat.json")) as f:
        cases = json.load(f)["filter_ramp_taps"]
    for case in cases:
        nt = case["ntaps"]
        taps = (np.arange(nt, dtype=np.float32) / np.float32(1000.0)).astype(np.float32)
        n = 8192  # the CLI's default block size
        x = np.full(n + nt - 1, complex(*case["input"]), np.complex64)
        y = np.empty(n, np.complex64)
        blk = gpu.clFilter(*GPU_ARGS, 1, taps, 1, 0, use_time)
        assert blk.work(n, [x], [y]) == n
        want = complex(*case["expect"])
        assert np.abs(y - want).max() <= TOL * abs(want), (nt, use_time)


@pytest.mark.parametrize("use_time", [False, True])
def test_independent_scipy_cases(gpu, use_time):
    """Against scipy.signal (firwin taps through upfirdn at decimations 1, 2, 3, 10, real and complex taps; lfilter; fftconvolve for 3001 taps)
    -- implementations that are not this repository's (tests/golden/gen_golden.py::independent_golden), both kernels of both blocks."""
    g = golden("independent_golden.npz")
    x, taps, ctaps, lt = g["fa_x"], g["fa_taps"], g["fa_ctaps"], g["fa_long_taps"]
    for d in (1, 2, 3, 10):
        n = x.size // d
        y = np.empty(n, np.complex64)
        assert gpu.clFilter(*GPU_ARGS, d, taps, 1, 0, use_time).work(n, [_hist(x, 65)], [y]) == n
        assert relerr(y, g["fa_y_d%d" % d]) <= TOL, d
        gpu.clComplexFilter(*GPU_ARGS, d, ctaps, 1, 0, use_time=use_time).work(n, [_hist(x, 65)], [y])
        assert relerr(y, g["fa_yc_d%d" % d]) <= TOL, d
    y = np.empty(x.size, np.complex64)
    gpu.clFilter(*GPU_ARGS, 1, taps, 1, 0, use_time).work(x.size, [_hist(x, 65)], [y])
    assert relerr(y, g["fa_y_lfilter"]) <= TOL
    gpu.clFilter(*GPU_ARGS, 1, lt, 1, 0, use_time).work(x.size, [_hist(x, 3001)], [y])
    assert relerr(y, g["fa_y_long"]) <= TOL


@pytest.mark.parametrize("ntaps,decim,ctaps", [(65, 16, False), (65, 9, False), (77, 15, True), (200, 25, False), (33, 33, False), (129, 10, True), (1000, 21, False)])
def test_lds_staged_decimators_agree(gpu, oracle, monkeypatch, ntaps, decim, ctaps):
    """k_fir_dec2 (round 6: 16-byte staging and reads; even decimations, and odd ones with half of the waves reading shifted by a sample) against the oracle
    and against k_fir_dec_lds (MI355_FIR_DEC2_OFF) on the device path, several tiles with a ragged last one."""
    import torch
    monkeypatch.setenv("MI355_FIR_DEC_KERNEL", "lds")
    rng = np.random.default_rng(ntaps * 7 + decim)
    nout = 40000 + 13
    xh = crandn(rng, nout * decim + ntaps - 1)
    if ctaps:
        taps = (crandn(rng, ntaps) / np.sqrt(ntaps)).astype(np.complex64)
        ref = oracle.fir_ccc(taps, xh, nout * decim)[::decim][:nout]
        blk = gpu.clComplexFilter(*GPU_ARGS, decim, taps, 1, 0, use_time=True)
    else:
        taps = (rng.standard_normal(ntaps) / np.sqrt(ntaps)).astype(np.float32)
        ref = oracle.fir_ccf(taps, xh, nout * decim)[::decim][:nout]
        blk = gpu.clFilter(*GPU_ARGS, decim, taps, 1, 0, True)
    x = torch.from_numpy(xh.view(np.float32).reshape(-1, 2)).cuda()
    y = torch.zeros(nout, 2, device="cuda")
    blk.work_device(nout, [x], [y])
    torch.cuda.synchronize()
    got = y.cpu().numpy().view(np.complex64).reshape(-1)
    assert relerr(got, ref) <= TOL
    monkeypatch.setenv("MI355_FIR_DEC2_OFF", "1")
    y2 = torch.zeros_like(y)
    blk.work_device(nout, [x], [y2])
    torch.cuda.synchronize()
    assert relerr(y2.cpu().numpy().view(np.complex64).reshape(-1), ref) <= TOL
    assert relerr(got, y2.cpu().numpy().view(np.complex64).reshape(-1)) <= 2e-6  # (same products, same order; the compiler may contract differently)
