"""GPU parity: clMathOp / clMathConst through the C ABI vs the oracle."""
import numpy as np
import pytest

from conftest import GPU_ARGS, crandn, relerr

pytestmark = pytest.mark.gpu

TOL = 1e-5  # north_star: float within 1e-5 relative; integer paths bit exact


def _mk(gpu, cls, dtype, *a):
    return cls(dtype, *GPU_ARGS, *a)


@pytest.mark.parametrize("n", [1, 2, 3, 5, 8192, 8193, 100003])
@pytest.mark.parametrize("op", ["MULTIPLY", "ADD", "SUBTRACT", "MULTIPLY_CONJUGATE"])
def test_mathop_complex(gpu, oracle, op, n):
    rng = np.random.default_rng(n)
    a, b = crandn(rng, n), crandn(rng, n)
    blk = _mk(gpu, gpu.clMathOp, gpu.DTYPE_COMPLEX, getattr(gpu, "MATHOP_" + op))
    c = np.empty_like(a)
    assert blk.work(n, [a, b], [c]) == n
    ref = oracle.mathop(oracle.DTYPE_COMPLEX, getattr(oracle, "OP_" + op), a, b)
    if op in ("ADD", "SUBTRACT"):
        assert np.array_equal(c, ref)
    else:
        assert relerr(c, ref) <= TOL


def test_mathop_reference_known_answer(gpu):
    # lib/test_clenabled.cc:1596-1600: 8192 x (1,0.5)*(1,0.5) = (0.75,1.0)
    a = np.full(8192, 1 + 0.5j, np.complex64)
    c = np.empty_like(a)
    _mk(gpu, gpu.clMathOp, gpu.DTYPE_COMPLEX, gpu.MATHOP_MULTIPLY).work(8192, [a, a], [c])
    assert np.all(c == np.complex64(0.75 + 1j))


@pytest.mark.parametrize("op", ["MULTIPLY", "ADD", "SUBTRACT"])
def test_mathop_int_bit_exact_wraparound(gpu, oracle, op):
    rng = np.random.default_rng(5)
    n = 70001
    a = rng.integers(-2**31, 2**31, n, dtype=np.int64).astype(np.int32)
    b = rng.integers(-2**31, 2**31, n, dtype=np.int64).astype(np.int32)
    c = np.empty_like(a)
    _mk(gpu, gpu.clMathOp, gpu.DTYPE_INT, getattr(gpu, "MATHOP_" + op)).work(n, [a, b], [c])
    assert np.array_equal(c, oracle.mathop(oracle.DTYPE_INT, getattr(oracle, "OP_" + op), a, b))


@pytest.mark.parametrize("op", ["MULTIPLY", "ADD", "SUBTRACT"])
def test_mathop_float(gpu, oracle, op):
    rng = np.random.default_rng(6)
    n = 12347
    a, b = rng.standard_normal(n).astype(np.float32), rng.standard_normal(n).astype(np.float32)
    c = np.empty_like(a)
    _mk(gpu, gpu.clMathOp, gpu.DTYPE_FLOAT, getattr(gpu, "MATHOP_" + op)).work(n, [a, b], [c])
    assert np.array_equal(c, oracle.mathop(oracle.DTYPE_FLOAT, getattr(oracle, "OP_" + op), a, b))


@pytest.mark.parametrize("n", [1, 7, 8192, 65537])
@pytest.mark.parametrize("op", ["MULTIPLY", "ADD", "SUBTRACT", "COMPLEX_CONJUGATE", "EMPTY_W_COPY"])
def test_mathconst_complex(gpu, oracle, op, n):
    rng = np.random.default_rng(n + 1)
    a = crandn(rng, n)
    blk = gpu.clMathConst(gpu.DTYPE_COMPLEX, *GPU_ARGS, 2.5, getattr(gpu, "MATHOP_" + op))
    c = np.empty_like(a)
    blk.work(n, [a], [c])
    oop = {"COMPLEX_CONJUGATE": "CONJUGATE"}.get(op, op)
    assert np.array_equal(c, oracle.mathconst(oracle.DTYPE_COMPLEX, getattr(oracle, "OP_" + oop), 2.5, a))


def test_mathconst_set_k_and_known_answer(gpu):
    blk = gpu.clMathConst(gpu.DTYPE_COMPLEX, *GPU_ARGS, 2.0, gpu.MATHOP_MULTIPLY)
    a = np.full(8192, 1 + 0.5j, np.complex64)
    c = np.empty_like(a)
    blk.work(8192, [a], [c])
    assert np.all(c == np.complex64(2 + 1j))  # lib/test_clenabled.cc:1351-1356
    assert blk.k() == 2.0
    blk.set_k(-3.0)
    assert blk.k() == -3.0
    blk.work(8192, [a], [c])
    assert np.all(c == np.complex64(-3 - 1.5j))


def test_mathconst_int_and_float(gpu, oracle):
    rng = np.random.default_rng(9)
    ia = rng.integers(-2**31, 2**31, 9999, dtype=np.int64).astype(np.int32)
    for op in ("MULTIPLY", "ADD", "SUBTRACT"):
        c = np.empty_like(ia)
        gpu.clMathConst(gpu.DTYPE_INT, *GPU_ARGS, 7.0, getattr(gpu, "MATHOP_" + op)).work(ia.size, [ia], [c])
        assert np.array_equal(c, oracle.mathconst(oracle.DTYPE_INT, getattr(oracle, "OP_" + op), 7.0, ia))
    fa = rng.standard_normal(4097).astype(np.float32)
    c = np.empty_like(fa)
    gpu.clMathConst(gpu.DTYPE_FLOAT, *GPU_ARGS, 0.3, gpu.MATHOP_MULTIPLY).work(fa.size, [fa], [c])
    assert np.array_equal(c, oracle.mathconst(oracle.DTYPE_FLOAT, oracle.OP_MULTIPLY, 0.3, fa))


def test_host_path_multi_chunk_pipeline(gpu, oracle):
    """> 8 MiB per input: exercises the double-buffered staging across several chunks."""
    rng = np.random.default_rng(11)
    n = (3 << 20) + 12345  # 3 chunks of 1 Mi complex items + ragged tail
    a, b = crandn(rng, n), crandn(rng, n)
    c = np.empty_like(a)
    gpu.clMathOp(gpu.DTYPE_COMPLEX, *GPU_ARGS, gpu.MATHOP_MULTIPLY).work(n, [a, b], [c])
    assert relerr(c, oracle.mathop(1, oracle.OP_MULTIPLY, a, b)) <= TOL
    gpu.clMathConst(gpu.DTYPE_COMPLEX, *GPU_ARGS, 1.5, gpu.MATHOP_ADD).work(n, [a], [c])
    assert np.array_equal(c, oracle.mathconst(1, oracle.OP_ADD, 1.5, a))


def test_device_resident_path_full_size(gpu, oracle):
    """Device path on torch's stream at a size well past the caches, checked by a
    size-independent property (a*b then *conj(b) scales a by |b|^2) plus a sampled oracle check."""
    import torch
    n = 1 << 24
    g = torch.Generator(device="cuda").manual_seed(3)
    a = torch.randn(n, 2, device="cuda", generator=g)
    b = torch.randn(n, 2, device="cuda", generator=g)
    c = torch.empty_like(a)
    d = torch.empty_like(a)
    mul = gpu.clMathOp(gpu.DTYPE_COMPLEX, *GPU_ARGS, gpu.MATHOP_MULTIPLY)
    mulc = gpu.clMathOp(gpu.DTYPE_COMPLEX, *GPU_ARGS, gpu.MATHOP_MULTIPLY_CONJUGATE)
    mul.work_device(n, [a, b], [c])
    mulc.work_device(n, [c, b], [d])
    torch.cuda.synchronize()
    b2 = (b * b).sum(dim=1, keepdim=True)
    assert torch.allclose(d, a * b2, rtol=1e-5, atol=1e-5)
    sl = slice(12345, 12345 + 4096)
    an = a[sl].cpu().numpy().view(np.complex64).reshape(-1)
    bn = b[sl].cpu().numpy().view(np.complex64).reshape(-1)
    cn = c[sl].cpu().numpy().view(np.complex64).reshape(-1)
    assert relerr(cn, oracle.mathop(1, oracle.OP_MULTIPLY, an, bn)) <= TOL


def test_zero_items_is_a_noop(gpu):
    blk = gpu.clMathOp(gpu.DTYPE_COMPLEX, *GPU_ARGS, gpu.MATHOP_ADD)
    e = np.empty(0, np.complex64)
    assert blk.work(0, [e, e], [e]) == 0
