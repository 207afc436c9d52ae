"""GPU parity beyond 4 GiB per buffer: every kernel addresses its data with 32-bit offsets from a 64-bit group base (or through
raw buffers that are rebuilt per group), so buffers whose byte size passes 2^31 and 2^32 are the cases where that arithmetic
can go wrong.  288 GB of HBM make such calls ordinary on this device.  Each test runs one device-resident call over more than
4 GiB and compares windows around the 2 GiB and 4 GiB byte offsets (and the ragged end) with the oracle / an independent
float64 evaluation."""
import numpy as np
import pytest

from conftest import GPU_ARGS, relerr

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _c64(t):
    return t.cpu().numpy().view(np.complex64).reshape(-1)


def test_fft_4096_over_4gib(gpu):
    import torch
    N, frames = 4096, (1 << 17) + 3  # 4.0 GiB + 3 frames in, same out
    win = np.blackman(N).astype(np.float32)
    g = torch.Generator(device="cuda").manual_seed(11)
    x = torch.randn(frames * N, 2, device="cuda", generator=g)
    y = torch.empty_like(x)
    blk = gpu.clFFT(N, gpu.CLFFT_FORWARD, win, 1, *GPU_ARGS, 0, 1, True)
    blk.work_device(frames, [x], [y])
    torch.cuda.synchronize()
    w = torch.from_numpy(win.astype(np.float64)).cuda()
    for fr in (0, (1 << 16) - 1, 1 << 16, (1 << 17) - 1, 1 << 17, frames - 1):
        xs = torch.view_as_complex(x[fr * N:(fr + 1) * N].double().contiguous())
        ref = torch.fft.fftshift(torch.fft.fft(xs * w)).cpu().numpy()
        assert relerr(_c64(y[fr * N:(fr + 1) * N]), ref) <= TOL, fr
    del x, y
    torch.cuda.empty_cache()


@pytest.mark.parametrize("use_time", [False, True])
def test_filter_over_4gib(gpu, oracle, use_time):
    import torch
    taps = oracle.firdes_low_pass(1.0, 10e6, 1e6, 372000.0)
    n = (1 << 29) + 4321
    g = torch.Generator(device="cuda").manual_seed(12)
    x = torch.randn(n + 64, 2, device="cuda", generator=g)
    y = torch.empty(n, 2, device="cuda")
    gpu.clFilter(*GPU_ARGS, 1, taps, 1, 0, use_time).work_device(n, [x], [y])
    torch.cuda.synchronize()
    for o0 in (0, (1 << 28) - 1500, (1 << 29) - 1500, n - 3000):
        xs = _c64(x[o0:o0 + 3000 + 64])
        assert relerr(_c64(y[o0:o0 + 3000]), oracle.fir_ccf(taps, xs, 3000)) <= TOL, o0
    del x, y
    torch.cuda.empty_cache()


def test_mathop_over_4gib(gpu):
    import torch
    n = (1 << 29) + 5
    g = torch.Generator(device="cuda").manual_seed(13)
    a = torch.randn(n, 2, device="cuda", generator=g)
    b = torch.randn(n, 2, device="cuda", generator=g)
    c = torch.empty_like(a)
    gpu.clMathOp(1, *GPU_ARGS, gpu.MATHOP_MULTIPLY).work_device(n, [a, b], [c])
    torch.cuda.synchronize()
    for o0 in (0, (1 << 28) - 100, (1 << 29) - 100, n - 200):
        sa, sb = _c64(a[o0:o0 + 200]).astype(np.complex128), _c64(b[o0:o0 + 200]).astype(np.complex128)
        assert relerr(_c64(c[o0:o0 + 200]), sa * sb) <= 1e-6, o0
    del a, b, c
    torch.cuda.empty_cache()


def test_pfb_over_4gib(gpu, oracle):
    import torch
    taps = np.concatenate([oracle.firdes_low_pass(1.0, 64.0, 0.5, 0.0753), [0.0]]).astype(np.float32)
    M, buf = 64, (1 << 29) + 64 * 100
    chmap = list(range(M))
    blk = gpu.clPolyphaseChannelizer(*GPU_ARGS, taps, buf, M, M, chmap)
    g = torch.Generator(device="cuda").manual_seed(14)
    x = torch.randn(blk.ninput(), 2, device="cuda", generator=g)
    y = torch.empty(blk.noutput(), 2, device="cuda")
    blk.work_device([x], [y])
    torch.cuda.synchronize()
    steps = 40
    for s0 in (0, (1 << 22) - 20, (1 << 23) - 20, buf // M - steps):  # step s0 consumes samples from s0*M on
        xs = _c64(x[s0 * M:s0 * M + M * steps + taps.size - M])
        ref = oracle.pfb(taps, M * steps, M, M, chmap, xs, f64=True)
        assert relerr(_c64(y[s0 * M:(s0 + steps) * M]), ref) <= TOL, s0
    del x, y
    torch.cuda.empty_cache()


def test_xengine_ichar_over_4gib_exact(gpu):
    """34 stations x 1024 channels x 65536 frames of int8 pairs = 4.25 GiB in ONE integration (65536 frames is the longest
    the int32 accumulators allow: 65536 * 2 * 127^2 < 2^31).  Integer arithmetic: the result must be bit-identical to
    float((double)S * kd * kd) with S from an independent float64 evaluation (exact at these magnitudes)."""
    import torch
    N, F, T = 34, 1024, 65536
    g = torch.Generator(device="cuda").manual_seed(15)
    x = torch.randint(-127, 128, (T, N, F, 1, 2), dtype=torch.int8, device="cuda", generator=g)
    assert x.numel() > (1 << 32)
    blk = gpu.clXEngine(*GPU_ARGS, False, gpu.DTYPE_BYTE, 1, N, 1, 0, F, T, [])
    out = torch.zeros(blk.get_output_buffer_size(), 2, device="cuda")
    blk.xcorrelate_device(x, out)
    S = torch.zeros(F, N, N, dtype=torch.complex128, device="cuda")
    for t0 in range(0, T, 2048):
        z = x[t0:t0 + 2048, :, :, 0, :].double()
        zc = torch.complex(z[..., 0], z[..., 1]).permute(2, 0, 1).contiguous()  # [f, t, s]
        S += torch.matmul(zc.transpose(1, 2), zc.conj())                        # [f, s1, s2] = sum_t x_s1 conj(x_s2)
    kd = 0.007874015748031496063
    s1, s2 = torch.tril_indices(N, N, device="cuda")                             # row-major lower triangle: k = s1(s1+1)/2 + s2
    tri = S[:, s1, s2]
    ref = torch.stack([(tri.real * kd * kd).float(), (tri.imag * kd * kd).float()], dim=-1)
    torch.cuda.synchronize()
    assert torch.equal(out.view(F, -1, 2), ref)
    del x, out, S
    torch.cuda.empty_cache()
