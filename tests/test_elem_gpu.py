"""Remaining elementwise family: oracle vs float64 formulas (CPU) and GPU parity."""
import numpy as np
import pytest

from conftest import GPU_ARGS, crandn, relerr

TOL = 1e-5


def _cases(rng, n):
    a = (np.abs(rng.standard_normal(n)) + 0.05).astype(np.float32)
    b = (np.abs(rng.standard_normal(n)) + 0.05).astype(np.float32)
    z = crandn(rng, n + 1)
    ph = (rng.uniform(-10, 10, n)).astype(np.float32)
    z64 = z.astype(np.complex128)
    return {
        1: ((a,), (2.5 * np.log10(a.astype(np.float64)) - 3.0,), (2.5, -3.0)),
        2: ((a, b), (np.abs(10 * np.log10((a / b).astype(np.float64)) + 1.0),), (10.0, 1.0)),
        3: ((z[:n],), (np.abs(z64[:n]),), (0, 0)),
        4: ((z[:n],), (np.angle(z64[:n]),), (0, 0)),
        5: ((z[:n],), (np.abs(z64[:n]), np.angle(z64[:n])), (0, 0)),
        6: ((a, ph), (a.astype(np.float64) * np.exp(1j * ph.astype(np.float64)),), (0, 0)),
        7: ((z,), (0.75 * np.angle(z64[1:] * np.conj(z64[:-1])),), (0.75, 0)),
    }


def test_oracle_matches_float64_formulas(oracle):
    rng = np.random.default_rng(42)
    n = 5001
    for kind, (ins, refs, (p0, p1)) in _cases(rng, n).items():
        outs = oracle.elem(kind, n, ins, p0, p1)
        for o, r in zip(outs, refs):
            assert relerr(o, r) < 2e-6, kind
    # quadrature demod of a pure tone = gain * angular step (the FM discriminator identity)
    w = 0.3
    tone = np.exp(1j * w * np.arange(1000)).astype(np.complex64)
    assert np.allclose(oracle.elem(oracle.ELEM_QUADDEMOD, 999, (tone,), 2.0)[0], 2.0 * w, atol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 63, 8192, 100001])
def test_gpu_parity_all_kinds(gpu, oracle, n):
    rng = np.random.default_rng(n)
    mk = {1: lambda: gpu.clLog(*GPU_ARGS, 2.5, -3.0), 2: lambda: gpu.clSNR(*GPU_ARGS, 10.0, 1.0),
          3: lambda: gpu.clComplexToMag(*GPU_ARGS), 4: lambda: gpu.clComplexToArg(*GPU_ARGS),
          5: lambda: gpu.clComplexToMagPhase(*GPU_ARGS), 6: lambda: gpu.clMagPhaseToComplex(*GPU_ARGS),
          7: lambda: gpu.clQuadratureDemod(0.75, *GPU_ARGS)}
    for kind, (ins, refs, (p0, p1)) in _cases(rng, n).items():
        blk = mk[kind]()
        assert blk.history() == (2 if kind == 7 else 1)
        outs = [np.empty(n, r.dtype if kind != 6 else np.complex64).astype(np.complex64 if kind == 6 else np.float32) for r in refs]
        assert blk.work(n, list(ins), outs) == n
        exp = oracle.elem(kind, n, ins, p0, p1)
        for o, e, r in zip(outs, exp, refs):
            assert relerr(o, e) <= TOL and relerr(o, r) <= TOL, kind


@pytest.mark.gpu
def test_gpu_device_path_and_errors(gpu):
    import torch
    n = 1 << 22
    z = torch.randn(n, 2, device="cuda")
    mag = torch.empty(n, device="cuda")
    gpu.clComplexToMag(*GPU_ARGS).work_device(n, [z], [mag])
    torch.cuda.synchronize()
    assert torch.allclose(mag, torch.linalg.vector_norm(z, dim=1), rtol=1e-6, atol=1e-6)
    blk = gpu.clQuadratureDemod(1.0, *GPU_ARGS)
    with pytest.raises(ValueError):
        blk.work(10, [np.zeros(10, np.complex64)], [np.empty(10, np.float32)])  # history item missing


@pytest.mark.gpu
def test_device_path_unaligned_pointers_and_tails(gpu, oracle):
    """The device path moves 4 items per thread through 16-byte accesses when every pointer is 16-byte aligned and one
    item per thread otherwise; both, plus the n % 4 tail, must give the same numbers."""
    import torch
    rng = np.random.default_rng(77)
    n = 10007
    c = crandn(rng, n + 3)
    blk = gpu.clComplexToMagPhase(*GPU_ARGS)
    ref_mag, ref_ph = oracle.elem(5, n, [c[:n]])
    dc = torch.from_numpy(c.view(np.float32).reshape(-1, 2)).cuda()
    for off in (0, 1):  # off = 1: the input starts 8 bytes into an allocation, outputs 4 bytes in
        mag = torch.empty(n + 1, device="cuda")
        ph = torch.empty(n + 1, device="cuda")
        cin = torch.from_numpy(c[:n + 1].view(np.float32).reshape(-1, 2)).cuda() if off == 0 else dc
        x = cin[off:off + n]
        if off:
            ref_mag, ref_ph = oracle.elem(5, n, [c[off:off + n]])
        blk.work_device(n, [x], [mag[off:off + n], ph[off:off + n]])
        torch.cuda.synchronize()
        assert relerr(mag[off:off + n].cpu().numpy(), ref_mag) <= 1e-5
        assert relerr(ph[off:off + n].cpu().numpy(), ref_ph) <= 1e-5
    q = gpu.clQuadratureDemod(0.7, *GPU_ARGS)
    (ref_q,) = oracle.elem(7, n, [c[:n + 1]], p0=0.7)
    out = torch.empty(n, device="cuda")
    q.work_device(n, [dc[:n + 1]], [out])
    torch.cuda.synchronize()
    assert relerr(out.cpu().numpy(), ref_q) <= 1e-5


@pytest.mark.gpu
def test_arg_kernels_round_like_float64_atan2(gpu):
    """clComplexToArg / clComplexToMagPhase / clQuadratureDemod evaluate atan2 in double like the reference
    (lib/clComplexToArg_impl.cc:145-147, lib/clQuadratureDemod_impl.cc:125-143): every output must be the float64 result rounded to
    float (one unit in the last place allowed for the rare value that sits on a rounding boundary), over 36 decades of dynamic range,
    all octants, the axes, signed zeros and non-finite arguments."""
    rng = np.random.default_rng(99)
    n = 1 << 20
    z = crandn(rng, n)
    z[: n // 4] *= (10.0 ** rng.uniform(-18, 18, n // 4)).astype(np.float32)
    sp = np.array([0, 1, -1, 1j, -1j, 1 + 1j, -1 + 1j, -1 - 1j, 1 - 1j, complex(0.0, -0.0), complex(-0.0, 0.0), complex(-1.0, -0.0),
                   complex(-1.0, 0.0), complex(np.inf, 1), complex(-np.inf, 1), complex(1, np.inf), complex(np.inf, np.inf), complex(np.nan, 1),
                   1e-30 + 1j, 1 + 1e-30j, 0.19891237 + 1j, 1 + 0.19891237j, 0.66817864 + 1j, 1 + 0.41421357j], dtype=np.complex64)
    z[-len(sp):] = sp
    ref = np.arctan2(z.imag.astype(np.float64), z.real.astype(np.float64))
    out = np.empty(n, np.float32)
    assert gpu.clComplexToArg(*GPU_ARGS).work(n, [z], [out]) == n
    r32 = ref.astype(np.float32)
    ok = (out == r32) | (np.isnan(out) & np.isnan(r32))
    ulp = np.abs(out[~ok].astype(np.float64) - ref[~ok]) / np.spacing(np.abs(r32[~ok])).astype(np.float64)
    assert np.count_nonzero(~ok) <= 2 and (ulp <= 0.5000001).all(), (np.count_nonzero(~ok), z[~ok][:4], out[~ok][:4], r32[~ok][:4])
    assert np.array_equal(np.signbit(out[-len(sp):]), np.signbit(r32[-len(sp):]))  # -0 -> -0 / -pi, +0 behind a negative real part -> +pi
    mag, ph = np.empty(n, np.float32), np.empty(n, np.float32)
    assert gpu.clComplexToMagPhase(*GPU_ARGS).work(n, [z], [mag, ph]) == n
    assert np.array_equal(ph, out, equal_nan=True)
    # quadrature demodulator: gain * atan2 of a[i+1] conj(a[i]) in double
    zq = crandn(rng, n + 1)
    d64 = zq[1:].astype(np.complex128) * np.conj(zq[:-1].astype(np.complex128))
    refq = (np.float64(np.float32(0.75)) * np.arctan2(d64.imag, d64.real))
    oq = np.empty(n, np.float32)
    assert gpu.clQuadratureDemod(0.75, *GPU_ARGS).work(n, [zq], [oq]) == n
    bad = oq != refq.astype(np.float32)
    assert np.count_nonzero(bad) <= n // 10000 and np.abs(oq - refq).max() <= 4e-7, np.count_nonzero(bad)
