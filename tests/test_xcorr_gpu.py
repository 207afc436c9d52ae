"""GPU parity: clxcorrelate_fft_vcf through the C ABI vs the oracle and vs the float64 definition.
The reference holds no vectors for this block (parity unpinned); the anchors are the oracle's restatement of
lib/clxcorrelate_fft_vcf_impl.cc:886-935,1058-1143 and the circular cross-correlation definition."""
import numpy as np
import pytest

from conftest import GPU_ARGS, crandn, golden, relerr

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _blk(gpu, n, nin, itype):
    ocl, sel, plat, dev = GPU_ARGS
    return gpu.clxcorrelate_fft_vcf(n, nin, ocl, sel, plat, dev, itype)


@pytest.mark.parametrize("n", [16, 32, 64, 128, 256, 512, 1024, 2048, 4096])
@pytest.mark.parametrize("itype", [1, 2])
def test_vs_oracle_all_sizes(gpu, oracle, n, itype):
    rng = np.random.default_rng(n + itype)
    nframes, nin = max(3, 9000 // n), 3   # more frames than one workgroup pass for small n, ragged last group
    ins = [crandn(rng, nframes * n) for _ in range(nin)]
    outs = [np.empty(nframes * n, np.float32) for _ in range(nin - 1)]
    blk = _blk(gpu, n, nin, itype)
    assert blk.work(nframes, ins, outs) == nframes
    ref = oracle.xcorr_fft(n, itype, ins, use_f64=True)
    for o, r in zip(outs, ref):
        assert relerr(o, r) <= TOL


def test_definition_and_peak_location(gpu):
    """Time-series input: a delayed copy of the reference peaks at lag -d, which the half swap puts at n/2 - d."""
    n, d = 1024, 37
    rng = np.random.default_rng(5)
    x0 = crandn(rng, n)
    x1 = np.roll(x0, d)
    blk = _blk(gpu, n, 2, 2)
    out = np.empty(n, np.float32)
    blk.work(1, [x0, x1], [out])
    assert int(np.argmax(out)) == (n // 2 - d) % n
    a, b = x0.astype(np.complex128), x1.astype(np.complex128)
    r = np.array([np.sum(np.roll(a, -m) * np.conj(b)) for m in range(n)]) * n
    assert relerr(out, np.fft.fftshift(np.abs(r)).astype(np.float32)) <= TOL


def test_many_inputs_and_device_path(gpu, oracle):
    import torch
    n, nframes, nin = 256, 1000, 6
    rng = np.random.default_rng(9)
    ins = [crandn(rng, nframes * n) for _ in range(nin)]
    blk = _blk(gpu, n, nin, 2)
    d_in = [torch.from_numpy(x.view(np.float32).reshape(-1, 2)).cuda() for x in ins]
    d_out = [torch.empty(nframes * n, device="cuda") for _ in range(nin - 1)]
    blk.work_device(nframes, d_in, d_out)
    torch.cuda.synchronize()
    ref = oracle.xcorr_fft(n, 2, ins, use_f64=True)
    for o, r in zip(d_out, ref):
        assert relerr(o.cpu().numpy(), r) <= TOL


def _np_xcorr(n, itype, ins):
    """The block's definition on numpy's float64 pocketfft (the oracle's O(N^2) DFT cannot run lengths that are not a power of two
    at these sizes); tied to the oracle at a small length in the test below."""
    x = [v.astype(np.complex128).reshape(-1, n) for v in ins]
    spec = x if itype == 1 else [np.fft.fft(v, axis=1) for v in x]
    outs = []
    for sp in spec[1:]:
        r = np.abs(np.fft.ifft(spec[0] * np.conj(sp), axis=1) * n)
        outs.append(np.concatenate([r[:, n // 2:], r[:, :n // 2]], axis=1).reshape(-1).astype(np.float32))
    return outs


# sizes outside the fused kernel's (powers of two 16 ... 4096): the reference's steps over the clFFT transforms -- powers of two below 16
# and above 4096 (one pass, two tile passes), 2-3-5-7 lengths (mixed radix, one and two passes), a length with a large prime factor (chirp-z)
@pytest.mark.parametrize("n", [2, 6, 14, 1000, 6000, 8192, 16384, 20000, 65536, 2 * 4099])
@pytest.mark.parametrize("itype", [1, 2])
def test_other_even_sizes(gpu, oracle, n, itype):
    import torch
    rng = np.random.default_rng(n + itype)
    nframes, nin = (7 if n < 5000 else 2), 3
    ins = [crandn(rng, nframes * n) for _ in range(nin)]
    outs = [np.empty(nframes * n, np.float32) for _ in range(nin - 1)]
    if n == 1000:  # the numpy reference against the oracle where the oracle can run
        for r, o in zip(_np_xcorr(n, itype, ins), oracle.xcorr_fft(n, itype, ins, use_f64=True)):
            assert relerr(r, o) <= 1e-6
    blk = _blk(gpu, n, nin, itype)
    assert blk.work(nframes, ins, outs) == nframes
    ref = _np_xcorr(n, itype, ins)
    for o, r in zip(outs, ref):
        assert relerr(o, r) <= TOL
    d_in = [torch.from_numpy(x.view(np.float32).reshape(-1, 2)).cuda() for x in ins]
    d_out = [torch.empty(nframes * n, device="cuda") for _ in range(nin - 1)]
    blk.work_device(nframes, d_in, d_out)
    torch.cuda.synchronize()
    for o, r in zip(d_out, ref):
        assert relerr(o.cpu().numpy(), r) <= TOL


def test_errors(gpu):
    with pytest.raises(gpu.Mi355Error):
        _blk(gpu, 1001, 2, 1)      # odd: the reference's half swap (vlen_2 = fftSize / 2) leaves the last output unwritten
    with pytest.raises(gpu.Mi355Error):
        _blk(gpu, (1 << 22) + 2, 2, 1)
    with pytest.raises(gpu.Mi355Error):
        _blk(gpu, 1024, 1, 1)      # needs a reference and one more input
    with pytest.raises(gpu.Mi355Error):
        _blk(gpu, 1024, 2, 3)      # input_type 1 or 2
    blk = _blk(gpu, 64, 2, 1)
    with pytest.raises(ValueError):
        blk.work(1, [np.zeros(64, np.complex64)], [np.zeros(64, np.float32)])


def test_independent_scipy_case(gpu):
    """Against scipy.signal.correlate (direct form, folded onto N circular lags) -- not this repository's arithmetic
    (tests/golden/gen_golden.py::independent_golden)."""
    g = golden("independent_golden.npz")
    ins = [g["xc_in%d" % i] for i in range(3)]
    outs = [np.empty(ins[0].size, np.float32) for _ in range(2)]
    assert _blk(gpu, 256, 3, 2).work(ins[0].size // 256, ins, outs) == ins[0].size // 256
    for o, s_ in zip(outs, (1, 2)):
        assert relerr(o, g["xc_out%d" % s_]) <= TOL
