"""The pybind11 module of the C++ block classes (host/python/bindings/python_bindings.cc) -- the `import clenabled` of an installed
GNU Radio build.  CPU: it imports, and every GRC make template's positional arguments bind to its constructors.  GPU: blocks
constructed positionally exactly as a GRC-generated flowgraph does, run through work() on numpy buffers, checked against the oracle."""
import glob
import inspect
import os
import re
import sys

import numpy as np
import pytest
import yaml

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "gr-clenabled_amd"))


def _mod():
    try:
        import clenabled_python
    except ImportError as exc:  # built by gr-clenabled_amd/host/Makefile when pybind11 is importable
        pytest.skip("clenabled_python not built: %s" % exc)
    return clenabled_python


def test_module_exposes_the_hot_path_blocks_and_enums():
    m = _mod()
    for cls in ("clMathOp", "clMathConst", "clFFT", "clFilter", "clComplexFilter", "clPolyphaseChannelizer", "clXEngine"):
        assert inspect.isclass(getattr(m, cls)), cls
    assert (m.CLFFT_FORWARD, m.CLFFT_BACKWARD, m.DTYPE_COMPLEX, m.DTYPE_BYTE, m.DTYPE_PACKEDXY, m.MATHOP_MULTIPLY_CONJUGATE) == (-1, 1, 1, 5, 6, 5)
    assert hasattr(m.clMathConst, "set_k") and hasattr(m.clFilter, "set_taps2") and hasattr(m.clComplexFilter, "set_taps2")


def test_grc_make_templates_match_the_binding_arity():
    """Count the positional arguments of every make: template against the pybind signature (its docstring lists the parameters)."""
    m = _mod()
    from test_grc_yaml import calls_of
    for path in sorted(glob.glob(os.path.join(ROOT, "gr-clenabled_amd", "grc", "clenabled_*.block.yml"))):
        d = yaml.safe_load(open(path))
        for cls, args in calls_of(d["templates"]["make"]):
            sig = getattr(m, cls).__init__.__doc__.split("(", 1)[1].split(") -> None")[0]
            params = [q for q in sig.split(", ") if not q.startswith("self:")]
            required = sum(" = " not in q for q in params)
            assert required <= len(args) <= len(params), "%s: %d arguments, the binding takes %d..%d" % (
                os.path.basename(path), len(args), required, len(params))


@pytest.mark.gpu
def test_blocks_built_the_way_grc_builds_them(gpu, oracle):
    m = _mod()
    rng = np.random.default_rng(2)

    def crandn(n):
        return (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)

    # clenabled.clFFT(${fft_size},${fft_dir},${window},${type.datatype},${openCLPlatform},${devices},${platformId},${deviceId},${setDebug},${num_streams},${shift})
    w = oracle.window(oracle.WIN_BLACKMAN_HARRIS, 1024)
    fft = m.clFFT(1024, -1, list(map(float, w)), 1, 1, 2, 0, 0, 0, 1, True)
    x, y = crandn(4 * 1024), np.empty(4 * 1024, np.complex64)
    assert fft.work(4, [x], [y]) == 4
    ref = oracle.fft_block(1024, True, w, True, oracle.DTYPE_COMPLEX, x, f64=True)
    assert np.abs(y - ref).max() <= 1e-5 * np.abs(ref).max()
    # clenabled.clMathOp(${type.datatype},${openCLPlatform},${devices},${platformId},${deviceId},1,${setDebug})
    a, b, c = crandn(8192), crandn(8192), np.empty(8192, np.complex64)
    assert m.clMathOp(1, 1, 2, 0, 0, 1, 0).work(8192, [a, b], [c]) == 8192
    assert np.abs(c - a * b).max() <= 1e-5 * np.abs(a * b).max()
    # clenabled.clMathConst(..., ${const}, 1, ${setDebug}) and the set_k callback
    k = m.clMathConst(1, 1, 2, 0, 0, 2.5, 1, 0)
    k.work(8192, [a], [c])
    assert np.allclose(c, np.float32(2.5) * a) and k.k() == 2.5
    k.set_k(-1.0)
    k.work(8192, [a], [c])
    assert np.array_equal(c, -a)
    # clenabled.clFilter(${openCLPlatform},${devices},${platformId},${deviceId},${decimation},firdes.low_pass(...),1,${setDebug},${use_time})
    taps = oracle.firdes_low_pass(1.0, 10e6, 1e6, 372000.0)
    for use_time in (True, False):
        f = m.clFilter(1, 2, 0, 0, 1, list(map(float, taps)), 1, 0, use_time)
        xh, yf = crandn(4096 + 64), np.empty(4096, np.complex64)
        assert f.work(4096, [xh], [yf]) == 4096
        r = oracle.fir_ccf(taps, xh, 4096)
        assert np.abs(yf - r).max() <= 1e-5 * np.abs(r).max()
        assert np.allclose(f.taps(), taps)
    # clenabled.clPolyphaseChannelizer(${openCLPlatform}, ${devices}, ${platformId}, ${deviceId}, ${taps}, ${buf_items}, ${num_channels}, ${ninputs_per_iter}, ${chmap})
    M, buf = 8, 8 * 64
    pt = rng.standard_normal(M * 6).astype(np.float32)
    p = m.clPolyphaseChannelizer(1, 2, 0, 0, list(map(float, pt)), buf, M, M, list(range(M)))
    xin, yo = crandn(2 * buf + pt.size - M), np.empty(2 * buf, np.complex64)
    assert p.general_work(2 * buf, [xin], [yo]) == 2 * buf  # two output multiples in one call
    r = oracle.pfb(pt, buf, M, M, list(range(M)), xin[:buf + pt.size - M], f64=True)
    assert np.abs(yo[:buf] - r).max() <= 1e-5 * np.abs(r).max()


def test_module_exposes_the_widened_blocks():
    m = _mod()
    for cls in ("clLog", "clSNR", "clComplexToMag", "clComplexToArg", "clComplexToMagPhase", "clMagPhaseToComplex", "clQuadratureDemod",
                "clxcorrelate_fft_vcf"):
        assert inspect.isclass(getattr(m, cls)), cls


@pytest.mark.gpu
def test_widened_blocks_built_the_way_grc_builds_them(gpu, oracle):
    """The elementwise family and the reference correlator through the pybind classes, positional arguments in the
    order of their GRC make templates, against the oracle."""
    m = _mod()
    rng = np.random.default_rng(5)
    n = 8192
    z = (rng.standard_normal(n + 1) + 1j * rng.standard_normal(n + 1)).astype(np.complex64)
    a = (np.abs(rng.standard_normal(n)) + 0.1).astype(np.float32)
    b = (np.abs(rng.standard_normal(n)) + 0.1).astype(np.float32)
    ph = rng.uniform(-3, 3, n).astype(np.float32)
    # clenabled.clLog(${openCLPlatform},${devices},${platformId},${deviceId},${n_val},${k_val},${setDebug}) etc.
    cases = [
        (m.clLog(1, 2, 0, 0, 2.5, -3.0, 0), oracle.ELEM_LOG10, (a,), (np.float32,), (2.5, -3.0)),
        (m.clSNR(1, 2, 0, 0, 10.0, 1.0, 0), oracle.ELEM_SNR, (a, b), (np.float32,), (10.0, 1.0)),
        (m.clComplexToMag(1, 2, 0, 0, 0), oracle.ELEM_C2MAG, (z[:n],), (np.float32,), (0, 0)),
        (m.clComplexToArg(1, 2, 0, 0, 0), oracle.ELEM_C2ARG, (z[:n],), (np.float32,), (0, 0)),
        (m.clComplexToMagPhase(1, 2, 0, 0, 0), oracle.ELEM_C2MAGPHASE, (z[:n],), (np.float32, np.float32), (0, 0)),
        (m.clMagPhaseToComplex(1, 2, 0, 0, 0), oracle.ELEM_MAGPHASE2C, (a, ph), (np.complex64,), (0, 0)),
        (m.clQuadratureDemod(0.75, 1, 2, 0, 0, 0), oracle.ELEM_QUADDEMOD, (z,), (np.float32,), (0.75, 0)),
    ]
    for blk, kind, ins, out_t, (p0, p1) in cases:
        outs = [np.empty(n, t) for t in out_t]
        assert blk.work(n, list(ins), outs) == n
        for o, e in zip(outs, oracle.elem(kind, n, ins, p0, p1)):
            assert np.abs(o - e).max() <= 2e-6 * max(1.0, np.abs(e).max()), kind
    # clenabled.clxcorrelate_fft_vcf(${vec_len},${num_inputs},${openCLPlatform},${devices},${platformId},${deviceId},${input_type})
    N, nin, fr = 256, 3, 5
    xs = [(rng.standard_normal(fr * N) + 1j * rng.standard_normal(fr * N)).astype(np.complex64) for _ in range(nin)]
    xc = m.clxcorrelate_fft_vcf(N, nin, 1, 2, 0, 0, 2)
    ys = [np.empty(fr * N, np.float32) for _ in range(nin - 1)]
    assert xc.work(fr, xs, ys) == fr
    ref = oracle.xcorr_fft(N, 2, xs, use_f64=True)
    for y, r in zip(ys, ref):
        assert np.abs(y - np.asarray(r, np.float32).reshape(-1)).max() <= 2e-5 * np.abs(r).max()


def test_clfilter_use_time_defaults_to_the_reference_value():
    """include/clenabled/clFilter.h:32,53 / python/bindings/clFilter_python.cc:48: use_time = DEFAULT_USE_TIME_DOMAIN_SETTING = false."""
    m = _mod()
    assert "use_time: bool = False" in m.clFilter.__init__.__doc__


@pytest.mark.gpu
def test_work_refuses_buffers_smaller_than_the_call(gpu):
    """The scheduler guarantees buffer sizes; a Python caller does not: undersized, strided or read-only buffers raise before any
    pointer reaches work()."""
    m = _mod()
    a, b, c = (np.zeros(8192, np.complex64) for _ in range(3))
    blk = m.clMathOp(1, 1, 2, 0, 0, 1, 0)
    with pytest.raises(ValueError):
        blk.work(8192, [a, b[:100]], [c])
    with pytest.raises(ValueError):
        blk.work(8192, [a, b], [c[:8000]])
    with pytest.raises(ValueError):
        blk.work(4096, [a[::2], b], [c])  # strided view
    ro = np.zeros(8192, np.complex64)
    ro.setflags(write=False)
    with pytest.raises(ValueError):
        blk.work(8192, [a, b], [ro])
    f = m.clFilter(1, 2, 0, 0, 2, [0.1] * 65, 1, 0)  # decimation 2, history 65 -> 4096 outputs need 8192 + 64 inputs
    x, y = np.zeros(8192 + 64, np.complex64), np.zeros(4096, np.complex64)
    assert f.work(4096, [x], [y]) == 4096
    with pytest.raises(ValueError):
        f.work(4096, [x[:8192]], [y])
    p = m.clPolyphaseChannelizer(1, 2, 0, 0, [0.1] * 48, 512, 8, 8, list(range(8)))
    with pytest.raises(ValueError):
        p.general_work(512, [np.zeros(512, np.complex64)], [np.zeros(512, np.complex64)])  # forecast() asks for 512 + 40


@pytest.mark.gpu
def test_clxengine_over_several_ranks_through_the_binding(gpu, oracle):
    """clXEngine.set_shard_devices (not in the reference: several devices behind ONE block, the corner turn inside the C ABI): frames fed through
    general_work() the way a flowgraph does, four ranks on the one device, one integration delivered; the block without it delivers the same."""
    m = _mod()
    N, F, T = 8, 64, 32
    rng = np.random.default_rng(9)
    streams = [rng.integers(-128, 128, size=(T, F, 2), dtype=np.int64).astype(np.int8) for _ in range(N)]
    for devs in ([0, 0, 0, 0], []):
        xe = m.clXEngine(1, 2, 0, 0, False, 5, 1, N, 1, 0, F, T, [])  # (data_type 5 = DTYPE_BYTE: IChar)
        if devs:
            xe.set_shard_devices(devs)
            assert xe.shard_devices() == 4
        else:
            assert xe.shard_devices() == 1
        assert xe.general_work(T, [s.reshape(-1).view(np.int8) for s in streams], []) == T
        xe.stop()
        assert xe.integrations_delivered() == 1
