"""GPU parity: clXEngine through the C ABI vs the oracle.  The reference holds no vectors
for this block; since round 4 the oracle and this path are pinned by fixtures computed with numpy.einsum / scipy.signal.correlate
(tests/golden/independent_golden.npz); further anchors are exact integer sums (golden fixtures),
closed-form cases and the oracle's restatement of the kernel text.
Integer paths (IChar, packed 4-bit sums) are required BIT EXACT against the oracle's exact mode."""
import os
import numpy as np
import pytest

from conftest import GPU_ARGS, crandn, golden, relerr

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _xe(gpu, dtype, npol, N, F, T):
    return gpu.clXEngine(*GPU_ARGS, False, dtype, npol, N, gpu.CLXCORR_TRIANGULAR_ORDER, 0, F, T, [])


def test_golden_exact_integer_sums(gpu):
    g = golden("xengine_golden.npz")
    N, F, T = (int(v) for v in g["cfg"])
    sc = 0.007874015748031496063
    for npol in (1, 2):
        blk = _xe(gpu, gpu.DTYPE_BYTE, npol, N, F, T)
        out = np.empty(blk.get_output_buffer_size(), np.complex64)
        blk.xcorrelate(g["i8_p%d_x" % npol], out)
        ref = ((g["i8_p%d_sum_re" % npol] * sc * sc) + 1j * (g["i8_p%d_sum_im" % npol] * sc * sc)).astype(np.complex64)
        assert np.array_equal(out, ref)  # bit exact
    blk = _xe(gpu, gpu.DTYPE_COMPLEX, 2, N, F, T)
    out = np.empty(blk.get_output_buffer_size(), np.complex64)
    blk.xcorrelate(g["cf_p2_x"], out)
    assert relerr(out, g["cf_p2_y"]) <= TOL
    blk = _xe(gpu, gpu.DTYPE_PACKEDXY, 2, N, F, T)
    out = np.empty(blk.get_output_buffer_size(), np.complex64)
    blk.xcorrelate(g["p4_x"], out)
    assert relerr(out, g["p4_y"]) <= TOL  # includes the code-8 -> 0 LUT quirk (lib/clXEngine_impl.cc:833)


def test_independent_numpy_scipy_cases(gpu):
    """The HIP path against numpy.einsum's exact integer sums in numpy.tril_indices order (bit for bit), the same contraction in
    complex128 on the scaled samples, and scipy.signal.correlate's zero lag for single baselines
    (tests/golden/gen_golden.py::independent_golden): data no author of the kernels or of the oracle wrote the arithmetic for."""
    g = golden("independent_golden.npz")
    kd = 0.007874015748031496063
    for c in ("xa", "xb", "xc"):
        N, F, T, npol = (int(v) for v in g[c + "_cfg"])
        blk = _xe(gpu, gpu.DTYPE_BYTE, npol, N, F, T)
        out = np.empty(blk.get_output_buffer_size(), np.complex64)
        blk.xcorrelate(g[c + "_x"], out)
        ref = ((g[c + "_sum_re"].astype(np.float64) * kd * kd).astype(np.float32)
               + 1j * (g[c + "_sum_im"].astype(np.float64) * kd * kd).astype(np.float32)).astype(np.complex64)
        assert np.array_equal(out, ref), c
        assert relerr(out, g[c + "_y"]) <= 1e-6, c
        xf = (g[c + "_x"].astype(np.float32) / np.float32(127.0)).view(np.complex64)  # the complex-float path on the same samples
        blk = _xe(gpu, gpu.DTYPE_COMPLEX, npol, N, F, T)
        blk.xcorrelate(xf, out)
        assert relerr(out, g[c + "_y"]) <= TOL, c
    N, F, T, npol = (int(v) for v in g["xa_cfg"])
    blk = _xe(gpu, gpu.DTYPE_BYTE, npol, N, F, T)
    out = np.empty(blk.get_output_buffer_size(), np.complex64)
    blk.xcorrelate(g["xa_x"], out)
    v = out.reshape(F, N * (N + 1) // 2)
    for (s1, s2, f), want in zip(g["xa_picks"], g["xa_pick_vals"]):
        assert abs(v[f, s1 * (s1 + 1) // 2 + s2] - want) <= 1e-6 * max(1.0, abs(want))


@pytest.mark.parametrize("N,F,T,npol", [(2, 2, 1, 1), (3, 6, 5, 1), (16, 8, 64, 1), (17, 4, 65, 1), (33, 10, 130, 2), (5, 1, 9, 1), (12, 7, 70, 1),
                                        (64, 16, 256, 1), (64, 4, 128, 2), (40, 6, 200, 2), (100, 2, 70, 1),
                                        # whole 128-byte input rows and <= 64 rows: the fused single-pass kernels (xengine_fused.hip), every
                                        # row-tile count, one and two polarisations, direct and time-split (partial sums) forms
                                        (64, 64, 64, 1), (20, 128, 96, 1), (5, 64, 32, 1), (33, 64, 256, 1), (48, 192, 128, 1), (64, 128, 512, 1),
                                        (16, 32, 32, 2), (32, 64, 128, 2), (7, 96, 64, 2), (24, 32, 1024, 2),
                                        # ... and integrations that are not whole K blocks of 32 frames (the frames past the end read as zeros)
                                        (64, 64, 100, 1), (20, 128, 1000, 1), (32, 64, 33, 2), (5, 64, 1, 1), (48, 64, 2047, 1), (16, 32, 31, 2),
                                        # ... and rows of whole 16-byte pieces that end inside a 128-byte line (the missing pieces read as zeros)
                                        (64, 1000, 64, 1), (20, 72, 96, 1), (33, 40, 100, 1), (16, 20, 64, 2), (64, 8, 1000, 1), (40, 200, 77, 1), (9, 7, 50, 1)])
def test_ichar_bit_exact_vs_oracle(gpu, oracle, N, F, T, npol):
    rng = np.random.default_rng(N * 1000 + T)
    x = rng.integers(-128, 128, size=T * N * F * npol * 2, dtype=np.int64).astype(np.int8)  # full range incl. -128
    blk = _xe(gpu, gpu.DTYPE_BYTE, npol, N, F, T)
    assert blk.get_output_buffer_size() == oracle.xengine_out_len(N, F, npol)
    out = np.empty(blk.get_output_buffer_size(), np.complex64)
    blk.xcorrelate(x, out)
    assert np.array_equal(out, oracle.xengine_ichar(N, F, npol, T, x, exact=True))
    # and the reference's own float arithmetic agrees within the budget
    assert relerr(out, oracle.xengine_ichar(N, F, npol, T, x, exact=False)) <= TOL


@pytest.mark.parametrize("N,F,T,npol", [(128, 64, 256, 1), (256, 16, 128, 1), (64, 64, 256, 2),  # 8 and 16 row tiles, whole 128-byte rows
                                        (130, 64, 128, 1), (200, 64, 64, 1), (96, 32, 192, 2), (81, 32, 100, 2),  # 9 -> 10, 13 -> 14, 12, 11 -> 12 row tiles
                                        (160, 10, 70, 1), (129, 3, 65, 1),                                        # rows that are not whole lines (slow corner turn)
                                        (256, 300, 64, 1)])                                                       # more channels than CUs: a run of channels per workgroup
def test_large_arrays_bit_exact_vs_oracle(gpu, oracle, N, F, T, npol):
    """More than 64 rows: corner turn + the persistent one-pass correlator (k_xe_corr_sb: a channel's whole triangle in one workgroup's
    registers, waves own tile rows w and NT-1-w).  Full int8 range; every padded row-tile count; two polarisations; ragged time."""
    rng = np.random.default_rng(N * 1000 + T + npol)
    x = rng.integers(-128, 128, size=T * N * F * npol * 2, dtype=np.int64).astype(np.int8)
    blk = _xe(gpu, gpu.DTYPE_BYTE, npol, N, F, T)
    out = np.empty(blk.get_output_buffer_size(), np.complex64)
    blk.xcorrelate(x, out)
    ref = oracle.xengine_ichar(N, F, npol, T, x, exact=True)
    assert np.array_equal(out, ref)
    if (N, F) == (130, 64):  # the accumulate form (pipeline integration) and a second integration through the same workspace
        x2 = rng.integers(-128, 128, size=x.size, dtype=np.int64).astype(np.int8)
        blk.xcorrelate(x2, out, accumulate=True)
        assert np.array_equal(out, oracle.xengine_ichar(N, F, npol, T, x2, exact=True, acc=ref))


def test_large_array_extremes_and_packed4(gpu, oracle):
    """Every sample (-128, -128) at 200 stations: the ~Q / row-sum form of the imaginary part at its limits (re = 2 * 128^2 * T, im = 0);
    packed 4-bit input at 160 rows (80 stations x 2) through the same correlator."""
    N, F, T = 200, 64, 4096
    blk = _xe(gpu, gpu.DTYPE_BYTE, 1, N, F, T)
    out = np.empty(blk.get_output_buffer_size(), np.complex64)
    blk.xcorrelate(np.full(T * N * F * 2, -128, np.int8), out)
    kd = 0.007874015748031496063
    assert np.array_equal(out, np.full(out.shape, np.float32(2.0 * 128 * 128 * T * kd * kd), np.complex64))
    N, F, T = 80, 64, 96
    rng = np.random.default_rng(3)
    x = rng.integers(0, 256, size=T * N * F * 2, dtype=np.int64).astype(np.uint8)
    blk = _xe(gpu, gpu.DTYPE_PACKEDXY, 2, N, F, T)
    out = np.empty(blk.get_output_buffer_size(), np.complex64)
    blk.xcorrelate(x, out)
    assert relerr(out, oracle.xengine_packed4(N, F, T, x)) <= TOL


@pytest.mark.parametrize("tsplit", ["1", "4", None])
def test_fused_full_range_extremes(gpu, oracle, monkeypatch, tsplit):
    """Every sample (-128, -128) over the longest integration: re = 2 * 128^2 * 65536 = 2^31 wraps the int32 accumulator of a single
    time range and must still come out exact; also the accumulate form on the fused path."""
    if tsplit is None:
        monkeypatch.delenv("MI355_XE_TSPLIT", raising=False)
    else:
        monkeypatch.setenv("MI355_XE_TSPLIT", tsplit)
    N, F, T = 3, 64, 65536
    x = np.full(T * N * F * 2, -128, np.int8)
    blk = _xe(gpu, gpu.DTYPE_BYTE, 1, N, F, T)
    out = np.empty(blk.get_output_buffer_size(), np.complex64)
    blk.xcorrelate(x, out)
    kd = 0.007874015748031496063
    assert np.array_equal(out, np.full(out.shape, np.float32(2147483648.0 * kd * kd), np.complex64))
    rng = np.random.default_rng(5)
    N, F, T = 40, 64, 128
    x = rng.integers(-128, 128, size=T * N * F * 2, dtype=np.int64).astype(np.int8)
    blk = _xe(gpu, gpu.DTYPE_BYTE, 1, N, F, T)
    out = np.empty(blk.get_output_buffer_size(), np.complex64)
    blk.xcorrelate(x, out)
    ref = oracle.xengine_ichar(N, F, 1, T, x, exact=True)
    assert np.array_equal(out, ref)
    blk.xcorrelate(x, out, True)
    assert np.array_equal(out, ref + ref)


@pytest.mark.gpu
@pytest.mark.parametrize("N,F,T,npol,tsplit", [(64, 64, 1024, 1, "4"), (33, 64, 4096, 1, "16"), (32, 64, 512, 2, "2"), (17, 128, 2048, 1, "8"), (64, 64, 2048, 1, "4")])
def test_fused_24_bit_partial_sums_at_their_limits(gpu, oracle, monkeypatch, N, F, T, npol, tsplit):
    """Time ranges of at most 256 steps travel as 24-bit partial sums (16-bit and 8-bit planes, value - 1).  The extremes of that
    form: every sample (-128, -128) makes re = 2^23 per range (the one value that needs the -1), alternating (-128, -128) /
    (-128, 127) stations make |im| = 255 * 128 * 256 per range; plus random data; all bit-exact against the oracle's exact mode and
    equal to the 32-bit form.  (T / ranges = 512 in the last case: stays 32-bit.)"""
    monkeypatch.setenv("MI355_XE_TSPLIT", tsplit)
    rng = np.random.default_rng(T + N)
    A = N * npol
    cases = []
    cases.append(np.full((T, A, F, 2), -128, np.int8))
    alt = np.full((T, A, F, 2), -128, np.int8)
    alt[:, 1::2, :, 1] = 127
    cases.append(alt)
    alt2 = alt.copy()
    alt2[:, :, F // 2:, 0] = 127  # I = 127 on half of the channels: the most negative re
    cases.append(alt2)
    cases.append(rng.integers(-128, 128, size=(T, A, F, 2), dtype=np.int64).astype(np.int8))
    blk = _xe(gpu, gpu.DTYPE_BYTE, npol, N, F, T)
    for x4 in cases:
        # input layout [t][station][chan][pol]{I,Q}
        x = np.ascontiguousarray(x4.reshape(T, N, npol, F, 2).transpose(0, 1, 3, 2, 4)).reshape(-1)
        out = np.empty(blk.get_output_buffer_size(), np.complex64)
        blk.xcorrelate(x, out)
        ref = oracle.xengine_ichar(N, F, npol, T, x, exact=True)
        assert np.array_equal(out, ref)
        monkeypatch.setenv("MI355_XE_NO_PACK24", "1")
        out32 = np.empty_like(out)
        blk.xcorrelate(x, out32)
        monkeypatch.delenv("MI355_XE_NO_PACK24")
        assert np.array_equal(out32, ref)


@pytest.mark.parametrize("N,F,T,npol,tsplit,giveup", [(64, 64, 256, 1, "4", False), (64, 128, 512, 1, "4", False), (50, 64, 1024, 1, "4", False),
                                                      (32, 64, 512, 2, "4", False), (25, 128, 256, 2, "4", False),  # the reduce-scatter tail (64 rows x 4 ranges)
                                                      (64, 128, 512, 1, "4", True), (32, 64, 256, 2, "4", True),   # ... with every bounded wait running out at once
                                                      (33, 128, 512, 1, "8", False), (32, 64, 128, 2, "2", False), (20, 64, 1024, 1, "16", False)])
def test_fused_in_launch_reduction(gpu, oracle, monkeypatch, N, F, T, npol, tsplit, giveup):
    """The four time ranges of a 64-row slice combined by the fused kernel's own tail (reduce-scatter: write-through 24-bit pieces, an
    arrival word per slice, bounded waits with the last arriver finishing what others gave up) instead of the second kernel: bit-exact,
    also over repeated launches on the same workspace and with accumulation.  Geometries the tail does not cover (other range counts,
    fewer row tiles) take the second kernel and must agree as well.  giveup: MI355_XE_DBG=512 makes every workgroup but the last of a
    slice give up at once, so the fallback path produces the whole matrix."""
    monkeypatch.setenv("MI355_XE_INKERNEL_REDUCE", "1")
    monkeypatch.setenv("MI355_XE_TSPLIT", tsplit)
    if giveup:
        monkeypatch.setenv("MI355_XE_DBG", "512")
    rng = np.random.default_rng(N + T)
    blk = _xe(gpu, gpu.DTYPE_BYTE, npol, N, F, T)
    out = np.empty(blk.get_output_buffer_size(), np.complex64)
    for rep in range(3):
        x = rng.integers(-128, 128, size=T * N * F * npol * 2, dtype=np.int64).astype(np.int8)
        ref = oracle.xengine_ichar(N, F, npol, T, x, exact=True)
        blk.xcorrelate(x, out)
        assert np.array_equal(out, ref), rep
    blk.xcorrelate(x, out, True)
    assert np.array_equal(out, ref + ref)


def test_in_launch_reduction_survives_mode_switches(gpu, oracle, monkeypatch):
    """One handle, one workspace: reduce-kernel launches, in-launch-reduction launches and a non-fused launch (unaligned input: its corner
    turn writes tiles over the workspace, counters included) in any order.  The counters' epoch advances only with launches that use
    them and is reset after the workspace was overwritten -- otherwise no workgroup reaches its target and `out` keeps stale values."""
    import torch
    monkeypatch.setenv("MI355_XE_TSPLIT", "4")
    N, F, T = 48, 64, 256
    rng = np.random.default_rng(11)
    blk = _xe(gpu, gpu.DTYPE_BYTE, 1, N, F, T)
    per = blk.get_output_buffer_size()

    def run(inkernel, misaligned=False):
        monkeypatch.setenv("MI355_XE_INKERNEL_REDUCE", "1" if inkernel else "0")
        x = rng.integers(-128, 128, size=T * N * F * 2, dtype=np.int64).astype(np.int8)
        buf = torch.zeros(x.size + 16, dtype=torch.int8, device="cuda")
        off = 4 if misaligned else 0  # 4-byte aligned only: not the fused path
        buf[off:off + x.size] = torch.from_numpy(x).cuda()
        out = torch.full((per, 2), 7.0, device="cuda")  # stale values must not survive
        blk.xcorrelate_device(buf[off:off + x.size], out)
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy().view(np.complex64).reshape(-1), oracle.xengine_ichar(N, F, 1, T, x, exact=True)), (inkernel, misaligned)

    for inkernel, mis in ((False, False), (False, False), (True, False), (False, False), (True, False), (True, True), (True, False), (False, True), (True, False)):
        run(inkernel, mis)


def test_closed_form_cases(gpu):
    N, F, T = 8, 6, 32
    blk = _xe(gpu, gpu.DTYPE_BYTE, 1, N, F, T)
    out = np.empty(blk.get_output_buffer_size(), np.complex64)
    x = np.zeros((T, N, F, 1, 2), np.int8)
    x[..., 0] = 127
    blk.xcorrelate(x.reshape(-1), out)
    assert np.allclose(out, T, rtol=1e-6)  # every sample (127,0) -> V = T
    x[:, 1::2, :, :, 0] = -127
    x[:, 1::2, :, :, 1] = 127  # odd antennas (-127, 127): sign / conjugation sense
    blk.xcorrelate(x.reshape(-1), out)
    v = out.reshape(F, N * (N + 1) // 2)
    for s1 in range(N):
        for s2 in range(s1 + 1):
            z1 = (-1 + 1j) if s1 % 2 else 1
            z2 = (-1 + 1j) if s2 % 2 else 1
            assert np.allclose(v[:, s1 * (s1 + 1) // 2 + s2], T * z1 * np.conj(z2), rtol=1e-6)
    blk = _xe(gpu, gpu.DTYPE_COMPLEX, 1, N, F, T)
    ph = np.exp(2j * np.pi * np.arange(T * F) / 37.0).reshape(T, 1, F).repeat(N, 1).astype(np.complex64)
    blk.xcorrelate(ph.reshape(-1), out)  # identical unit-modulus tone on every antenna (lib/test-clxengine.cc:225-247)
    assert np.allclose(out, T, atol=1e-3)


# the fp32 matrix-core path (row tiles 1, 2, 3->4, 4, 5->6, 8; rows padded on the device to whole 128-byte lines when F*npol % 16 != 0); more than 256 rows: the VALU kernel
@pytest.mark.parametrize("N,F,T,npol", [(4, 8, 16, 1), (9, 5, 33, 2), (16, 32, 64, 1), (16, 16, 64, 1), (20, 16, 50, 1),
                                        (40, 32, 33, 1), (64, 16, 100, 1), (33, 8, 130, 2), (64, 8, 40, 2), (70, 16, 20, 2),
                                        # rows <= 64 and F % 8 == 0: the fused kernel (1, 2, 3->4 and 4 row tiles, both polarisation counts)
                                        (8, 8, 20, 2), (16, 8, 70, 2), (24, 16, 130, 2), (32, 16, 64, 2), (64, 24, 1000, 1), (5, 8, 3, 1)])
def test_complex_float_vs_oracle(gpu, oracle, N, F, T, npol):
    rng = np.random.default_rng(N + T)
    x = crandn(rng, T * N * F * npol)
    blk = _xe(gpu, gpu.DTYPE_COMPLEX, npol, N, F, T)
    out = np.zeros(blk.get_output_buffer_size(), np.complex64)
    blk.xcorrelate(x, out)
    ref = oracle.xengine_cf32(N, F, npol, T, x)
    assert relerr(out, ref) <= TOL
    blk.xcorrelate(x, out, accumulate=True)
    assert relerr(out, ref + ref) <= TOL


# channel counts whose rows are not whole 128-byte lines are padded on the device for the matrix-core kernels (no output for the padding
# channels); few channels x long integrations run many time ranges; MI355_XE_CF32_NO_PAD keeps the vector-ALU kernel in the suite
@pytest.mark.parametrize("N,F,T,npol", [(50, 100, 200, 1), (20, 37, 128, 1), (64, 10, 4096, 1), (12, 50, 300, 2), (30, 7, 1000, 2), (64, 2, 16384, 1),
                                        (100, 20, 64, 1), (3, 1, 2, 1),
                                        # 129 ... 256 rows: the triangle's tile pairs split over several workgroups per channel (10, 12, 16 row tiles)
                                        (130, 16, 48, 1), (180, 8, 40, 1), (100, 16, 33, 2), (256, 16, 32, 1), (200, 5, 64, 1)])
def test_complex_float_ragged_rows_and_many_time_ranges(gpu, oracle, monkeypatch, N, F, T, npol):
    rng = np.random.default_rng(N * 3 + F + T)
    x = crandn(rng, T * N * F * npol)
    ref = oracle.xengine_cf32(N, F, npol, T, x)
    for mode in ("in place", "padded copy", "vector ALU"):
        if mode == "padded copy":
            monkeypatch.setenv("MI355_XE_CF32_PAD_COPY", "1")  # read per call: the fused kernel on a padded copy instead of the caller's rows
        if mode == "vector ALU":
            monkeypatch.setenv("MI355_XE_CF32_NO_PAD", "1")  # read when the block is made
        blk = _xe(gpu, gpu.DTYPE_COMPLEX, npol, N, F, T)
        guard = np.full(blk.get_output_buffer_size() + 64, 7 + 7j, np.complex64)  # nothing may be written behind the last real channel
        out = guard[:blk.get_output_buffer_size()]
        blk.xcorrelate(x, out)
        assert relerr(out, ref) <= TOL and np.all(guard[out.size:] == 7 + 7j)
        blk.xcorrelate(x, out, accumulate=True)
        assert relerr(out, ref + ref) <= TOL
        out2 = np.empty_like(ref)  # the double-buffered host pipeline pads in its own slot buffers
        blk.submit(x)
        blk.submit(x)
        blk.wait(out2)
        assert relerr(out2, ref) <= TOL
        blk.wait(out2)
        assert relerr(out2, ref) <= TOL


# the vector-ALU kernel on channels padded to whole lines (more than 256 rows, or MI355_XE_CF32_VALU): the padding channels have no output.
# The guard is on the DEVICE, right behind the matrix (round 3's advisor finding: the kernel wrote Fout .. F - 1 past the end).
@pytest.mark.parametrize("N,F,T,npol,valu", [(260, 10, 24, 1, False), (130, 5, 16, 2, False), (300, 100, 4, 1, False), (20, 37, 64, 1, True), (12, 50, 40, 2, True)])
def test_complex_float_padded_channels_vector_alu_stays_in_bounds(gpu, oracle, monkeypatch, N, F, T, npol, valu):
    import torch
    if valu:
        monkeypatch.setenv("MI355_XE_CF32_VALU", "1")
    rng = np.random.default_rng(N + F + T)
    x = crandn(rng, T * N * F * npol)
    ref = oracle.xengine_cf32(N, F, npol, T, x)
    blk = _xe(gpu, gpu.DTYPE_COMPLEX, npol, N, F, T)
    n = blk.get_output_buffer_size()
    guard = 1 << 16
    buf = torch.full((n + guard, 2), 7.0, device="cuda")
    xd = torch.from_numpy(x.view(np.float32)).cuda()
    blk.xcorrelate_device(xd, buf[:n])
    torch.cuda.synchronize()
    got = buf.cpu().numpy()
    assert relerr(got[:n].view(np.complex64).ravel(), ref) <= TOL
    assert np.all(got[n:] == 7.0)
    blk.xcorrelate_device(xd, buf[:n], accumulate=True)
    torch.cuda.synchronize()
    got = buf.cpu().numpy()
    assert relerr(got[:n].view(np.complex64).ravel(), ref + ref) <= TOL and np.all(got[n:] == 7.0)
    out = np.zeros(n + 64, np.complex64)  # and through the host path
    blk.xcorrelate(x, out[:n])
    assert relerr(out[:n], ref) <= TOL and np.all(out[n:] == 0)


@pytest.mark.parametrize("N,F,T", [(2, 2, 3), (5, 6, 70), (16, 8, 64), (6, 5, 40), (3, 1, 2)])  # odd channel counts: padded on the device
def test_packed4_vs_oracle(gpu, oracle, N, F, T):
    rng = np.random.default_rng(N * 7 + T)
    x = rng.integers(0, 256, size=T * N * F * 2, dtype=np.int64).astype(np.uint8)
    blk = _xe(gpu, gpu.DTYPE_PACKEDXY, 2, N, F, T)
    out = np.empty(blk.get_output_buffer_size(), np.complex64)
    blk.xcorrelate(x, out)
    assert relerr(out, oracle.xengine_packed4(N, F, T, x)) <= TOL


def test_pipeline_integration_accumulates(gpu, oracle):
    N, F, T = 6, 4, 48
    rng = np.random.default_rng(77)
    x1 = rng.integers(-127, 128, size=T * N * F * 2, dtype=np.int64).astype(np.int8)
    x2 = rng.integers(-127, 128, size=T * N * F * 2, dtype=np.int64).astype(np.int8)
    blk = _xe(gpu, gpu.DTYPE_BYTE, 1, N, F, T)
    out = np.zeros(blk.get_output_buffer_size(), np.complex64)  # zero-filled accumulator (:289-292)
    blk.xcorrelate(x1, out, accumulate=True)
    blk.xcorrelate(x2, out, accumulate=True)  # "+=" (:785-796)
    ref = oracle.xengine_ichar(N, F, 1, T, x1, exact=True)
    ref = oracle.xengine_ichar(N, F, 1, T, x2, exact=True, acc=ref)
    assert np.array_equal(out, ref)


def test_gather_matches_reference_layout(gpu, oracle):
    N, F, T = 3, 6, 4
    rng = np.random.default_rng(3)
    for dtype, npol, esz, nin in ((gpu.DTYPE_BYTE, 2, 2, 2 * N), (gpu.DTYPE_BYTE, 1, 2, N), (gpu.DTYPE_COMPLEX, 2, 8, 2 * N),
                                  (gpu.DTYPE_PACKEDXY, 2, 1, N)):
        per = F * esz * (2 if dtype == gpu.DTYPE_PACKEDXY else 1)
        ins = [rng.integers(-127, 128, size=T * per, dtype=np.int64).astype(np.int8) for _ in range(nin)]
        blk = _xe(gpu, dtype, npol, N, F, T)
        a = np.zeros(blk.input_bytes(), np.int8)
        b = np.zeros(blk.input_bytes(), np.int8)
        blk.gather(2, 0, ins, a)
        blk.gather(2, 2, [i[2 * per:] for i in ins], a)
        oracle.xengine_gather(dtype, N, F, npol, T, 0, ins, b)
        assert np.array_equal(a, b)


def test_baseline_config5_full_size(gpu, oracle):
    """64 antennas x 1024 channels x 1024 frames, IChar (BASELINE configs[4]), device resident.
    Checks: a slab of channels bit-exact vs the oracle, Hermitian self-consistency of the
    autocorrelations (imag == 0, real == sum |x|^2), and the accumulate path doubling the result."""
    import torch
    N, F, T = 64, 1024, 1024
    g = torch.Generator(device="cuda").manual_seed(42)
    x = torch.randint(-127, 128, (T, N, F, 2), dtype=torch.int8, device="cuda", generator=g)
    blk = _xe(gpu, gpu.DTYPE_BYTE, 1, N, F, T)
    out = torch.zeros(blk.get_output_buffer_size(), 2, device="cuda")
    blk.xcorrelate_device(x, out)
    torch.cuda.synchronize()
    nb = N * (N + 1) // 2
    v = out.view(F, nb, 2)
    diag = torch.tensor([s * (s + 1) // 2 + s for s in range(N)], device="cuda")
    auto = v[:, diag, :]
    assert torch.all(auto[..., 1] == 0)
    pw = (x.to(torch.int64) ** 2).sum(dim=(0, 3)).t().to(torch.float64) * (0.007874015748031496063 ** 2)  # [F][N]
    assert torch.allclose(auto[..., 0].to(torch.float64), pw, rtol=1e-6)
    fs = [0, 1, 511, 1022, 1023]
    xs = x[:, :, fs, :].contiguous().cpu().numpy()
    ref = oracle.xengine_ichar(N, len(fs), 1, T, xs.reshape(-1), exact=True).reshape(len(fs), nb)
    got = v[fs].cpu().numpy().view(np.complex64).reshape(len(fs), nb)
    assert np.array_equal(got, ref)
    blk.xcorrelate_device(x, out, accumulate=True)
    torch.cuda.synchronize()
    got2 = out.view(F, nb, 2)[fs].cpu().numpy().view(np.complex64).reshape(len(fs), nb)
    assert np.array_equal(got2, ref + ref)


def test_constructor_errors(gpu):
    with pytest.raises(IndexError):
        _xe(gpu, gpu.DTYPE_BYTE, 1, 1, 16, 16)  # lib/clXEngine_impl.cc:106-109
    with pytest.raises(gpu.Mi355Error):
        _xe(gpu, gpu.DTYPE_BYTE, 3, 4, 16, 16)


def test_async_double_buffered_submit_wait(gpu, oracle):
    """submit()/wait(): two integrations in flight, results in order, equal to the synchronous path."""
    N, F, T = 16, 32, 128
    rng = np.random.default_rng(21)
    xs = [rng.integers(-128, 128, size=T * N * F * 2, dtype=np.int64).astype(np.int8) for _ in range(5)]
    refs = [oracle.xengine_ichar(N, F, 1, T, x, exact=True) for x in xs]
    blk = _xe(gpu, gpu.DTYPE_BYTE, 1, N, F, T)
    out = np.empty(blk.get_output_buffer_size(), np.complex64)
    got = []
    blk.submit(xs[0])
    for i in range(1, 5):
        blk.submit(xs[i])          # second one in flight while the first computes
        assert blk.pending() == 2
        with pytest.raises(gpu.Mi355Error):
            blk.submit(xs[i])      # a third is refused, not queued silently
        blk.wait(out)
        got.append(out.copy())
    blk.wait(out)
    got.append(out.copy())
    assert blk.pending() == 0
    with pytest.raises(gpu.Mi355Error):
        blk.wait(out)
    for g, r in zip(got, refs):
        assert np.array_equal(g, r)
    acc = refs[0].copy()
    blk.submit(xs[1], accumulator=acc)  # pipeline integration through the async path
    blk.wait(out)
    assert np.array_equal(out, refs[0] + refs[1])


def test_zero_copy_acquire_submit(gpu, oracle):
    """acquire()/submit_acquired(): frames gathered straight into the pinned slot buffer; same results, same ordering
    rules as submit()/wait()."""
    N, F, T = 12, 16, 64
    rng = np.random.default_rng(33)
    xs = [rng.integers(-128, 128, size=T * N * F * 2, dtype=np.int64).astype(np.int8) for _ in range(4)]
    refs = [oracle.xengine_ichar(N, F, 1, T, x, exact=True) for x in xs]
    blk = _xe(gpu, gpu.DTYPE_BYTE, 1, N, F, T)
    out = np.empty(blk.get_output_buffer_size(), np.complex64)
    got = []
    for i, x in enumerate(xs):
        buf = blk.acquire()
        assert buf.size == blk.input_bytes()
        with pytest.raises(gpu.Mi355Error):
            blk.submit(x)              # a plain submit is refused while a frame buffer is handed out
        per = N * F * 2                # bytes per frame: gather frame by frame like work_processor does
        for t in range(T):
            buf[t * per:(t + 1) * per] = x[t * per:(t + 1) * per]
        blk.submit_acquired()
        if blk.pending() == 2:
            blk.wait(out)
            got.append(out.copy())
    while blk.pending():
        blk.wait(out)
        got.append(out.copy())
    assert len(got) == 4
    for g, r in zip(got, refs):
        assert np.array_equal(g, r)
    with pytest.raises(gpu.Mi355Error):
        blk.submit_acquired()          # nothing acquired


@pytest.mark.parametrize("N,F,T,npol,W", [(64, 128, 256, 1, 8), (64, 256, 128, 1, 2), (32, 64, 64, 2, 4), (16, 64, 96, 1, 4)])
def test_group_major_input_read_in_place(gpu, oracle, N, F, T, npol, W):
    """Multi-GPU corner turn (shard.py): the all-to-all receive buffer [group][t][station in group][chan][pol] goes to the kernel
    as it is.  Bit-exact against the reference layout and the oracle; the send-side packing kernel against the index arithmetic."""
    import torch
    rng = np.random.default_rng(N + W)
    full = rng.integers(-128, 128, size=(T, N, F, npol, 2), dtype=np.int64).astype(np.int8)
    Ng = N // W
    grouped = np.ascontiguousarray(full.reshape(T, W, Ng, F, npol, 2).transpose(1, 0, 2, 3, 4, 5))
    blk = _xe(gpu, gpu.DTYPE_BYTE, npol, N, F, T)
    ref = oracle.xengine_ichar(N, F, npol, T, full.reshape(-1), exact=True)
    xg = torch.from_numpy(grouped).cuda()
    out = torch.zeros(blk.get_output_buffer_size(), 2, device="cuda")
    blk.xcorrelate_device(xg, out, stations_per_group=Ng)
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy().view(np.complex64).reshape(-1), ref)
    # send side: [t][station][peer][chan slice] -> [peer][t][station][chan slice]
    x = torch.from_numpy(full).cuda()
    send = torch.empty_like(x)
    esz, Fw = npol * 2, F // W
    blk.pack3d_device(send, x, Fw * esz, T * N, W, F * esz, Fw * esz, Fw * esz, T * N * Fw * esz)
    torch.cuda.synchronize()
    want = full.reshape(T, N, W, Fw, npol, 2).transpose(2, 0, 1, 3, 4, 5)
    assert np.array_equal(send.cpu().numpy().reshape(W, T, N, Fw, npol, 2), want)


def test_group_major_input_refused_outside_the_fused_path(gpu):
    import torch
    blk = _xe(gpu, gpu.DTYPE_BYTE, 1, 80, 16, 32)  # more than 64 rows: not the fused path
    x = torch.zeros(32 * 80 * 16 * 2, dtype=torch.int8, device="cuda")
    out = torch.zeros(blk.get_output_buffer_size(), 2, device="cuda")
    with pytest.raises(gpu.Mi355Error):
        blk.xcorrelate_device(x, out, stations_per_group=4)
    blk = _xe(gpu, gpu.DTYPE_BYTE, 1, 8, 5, 32)   # 10-byte rows: not whole 16-byte pieces
    x = torch.zeros(32 * 8 * 5 * 2, dtype=torch.int8, device="cuda")
    out = torch.zeros(blk.get_output_buffer_size(), 2, device="cuda")
    with pytest.raises(gpu.Mi355Error):
        blk.xcorrelate_device(x, out, stations_per_group=4)
    # 32-byte rows (a quarter of a line) are inside the fused path since round 3: group-major input = the same matrix as the reference layout
    N, F, T, ng = 8, 16, 33, 4
    rng = np.random.default_rng(11)
    xr = rng.integers(-128, 128, size=(T, N, F, 2), dtype=np.int64).astype(np.int8)
    blk = _xe(gpu, gpu.DTYPE_BYTE, 1, N, F, T)
    a = torch.zeros(blk.get_output_buffer_size(), 2, device="cuda")
    b = torch.zeros_like(a)
    blk.xcorrelate_device(torch.from_numpy(xr).cuda(), a)
    xg = np.ascontiguousarray(xr.reshape(T, N // ng, ng, F, 2).transpose(1, 0, 2, 3, 4))  # [group][t][station in group][chan]
    blk.xcorrelate_device(torch.from_numpy(xg).cuda(), b, stations_per_group=ng)
    torch.cuda.synchronize()
    assert torch.equal(a, b)


@pytest.mark.parametrize("N,F,T,npol,nint,W", [(64, 128, 1024, 1, 1, 1), (64, 128, 1024, 1, 3, 1), (64, 128, 1024, 1, 8, 1),   # the per-rank problem of config 5
                                              (64, 128, 256, 1, 8, 8), (16, 64, 128, 2, 3, 4), (64, 128, 128, 1, 40, 1),     # group-major; enough windows for no time split
                                              (20, 64, 96, 1, 5, 1), (12, 7, 70, 1, 3, 1), (100, 64, 64, 1, 2, 1)])          # other tile counts; non-fused fallbacks
def test_batched_integration_windows_bit_exact(gpu, oracle, N, F, T, npol, nint, W):
    """mi355_xengine_xcorrelate_n_dev: nint integration windows per launch (reference: one window per pass of the worker thread's loop,
    lib/clXEngine_impl.cc:1234-1299).  Every window bit-exact against the oracle, in the reference layout and in the group-major layout
    one all-to-all over nint windows delivers; then once more with accumulate."""
    import torch
    rng = np.random.default_rng(N * 31 + nint)
    wins = rng.integers(-128, 128, size=(nint, T, N, F, npol, 2), dtype=np.int64).astype(np.int8)
    blk = _xe(gpu, gpu.DTYPE_BYTE, npol, N, F, T)
    per = blk.get_output_buffer_size()
    ref = np.concatenate([oracle.xengine_ichar(N, F, npol, T, wins[i].reshape(-1), exact=True) for i in range(nint)])
    out = torch.zeros(nint * per, 2, device="cuda")
    if W > 1:
        Ng = N // W  # [window][t][group][station] -> [group][window][t][station in group]
        x = torch.from_numpy(np.ascontiguousarray(wins.reshape(nint, T, W, Ng, F, npol, 2).transpose(2, 0, 1, 3, 4, 5, 6))).cuda()
        blk.xcorrelate_n_device(nint, x, out, stations_per_group=Ng)
    else:
        x = torch.from_numpy(wins).cuda()
        blk.xcorrelate_n_device(nint, x, out)
    torch.cuda.synchronize()
    got = out.cpu().numpy().view(np.complex64).reshape(-1)
    assert np.array_equal(got, ref)
    if W == 1:
        blk.xcorrelate_n_device(nint, x, out, accumulate=True)
        torch.cuda.synchronize()
        ref2 = np.concatenate([oracle.xengine_ichar(N, F, npol, T, wins[i].reshape(-1), exact=True, acc=ref[i * per:(i + 1) * per].copy()) for i in range(nint)])
        assert np.array_equal(out.cpu().numpy().view(np.complex64).reshape(-1), ref2)
    # a single-window call through the same handle afterwards (its own workspace) is unaffected
    one = torch.zeros(per, 2, device="cuda")
    blk.xcorrelate_device(torch.from_numpy(wins[nint - 1]).cuda(), one)
    torch.cuda.synchronize()
    assert np.array_equal(one.cpu().numpy().view(np.complex64).reshape(-1), ref[(nint - 1) * per:])


@pytest.mark.parametrize("N,F,T,npol,nint,W,shift", [(64, 512, 32, 1, 10, 1, 0), (20, 1024, 40, 1, 5, 1, 3), (32, 512, 32, 2, 5, 1, 5),
                                                    (64, 512, 32, 1, 9, 8, 2), (16, 2048, 16, 1, 3, 1, 7)])
def test_batched_more_units_than_cus_slow_lines_first(gpu, oracle, N, F, T, npol, nint, W, shift):
    """More whole-integration units than CUs in one launch: the units of the rows' slow 128-byte lines (address bits 7..9 == 3) get the lowest
    workgroup numbers (FuArgs::slow_first).  The permutation depends on the input's address class, so the input is placed at every shift of
    128 bytes tried here; every window bit-exact against the oracle, reference and group-major layout."""
    import torch
    rng = np.random.default_rng(N * 7 + nint + shift)
    wins = rng.integers(-128, 128, size=(nint, T, N, F, npol, 2), dtype=np.int64).astype(np.int8)
    blk = _xe(gpu, gpu.DTYPE_BYTE, npol, N, F, T)
    per = blk.get_output_buffer_size()
    ref = np.concatenate([oracle.xengine_ichar(N, F, npol, T, wins[i].reshape(-1), exact=True) for i in range(nint)])
    if W > 1:
        Ng = N // W
        host = np.ascontiguousarray(wins.reshape(nint, T, W, Ng, F, npol, 2).transpose(2, 0, 1, 3, 4, 5, 6)).reshape(-1)
    else:
        Ng = 0
        host = wins.reshape(-1)
    raw = torch.zeros(host.size + 1024, dtype=torch.int8, device="cuda")
    x = raw[128 * shift:128 * shift + host.size]
    x.copy_(torch.from_numpy(host))
    out = torch.zeros(nint * per, 2, device="cuda")
    for env in (None, "1"):  # second pass: the plain order, same results
        # (the order is the default only when the early touches of the slow lines are off: MI355_XE_SLOW_FIRST forces it beside them)
        os.environ["MI355_XE_NO_SLOW_FIRST" if env else "MI355_XE_SLOW_FIRST"] = "1"
        try:
            out.zero_()
            if W > 1: blk.xcorrelate_n_device(nint, x, out, stations_per_group=Ng)
            else: blk.xcorrelate_n_device(nint, x, out)
            torch.cuda.synchronize()
        finally:
            os.environ.pop("MI355_XE_NO_SLOW_FIRST", None)
            os.environ.pop("MI355_XE_SLOW_FIRST", None)
        assert np.array_equal(out.cpu().numpy().view(np.complex64).reshape(-1), ref)


def test_batched_group_major_refused_outside_the_fused_path(gpu):
    import torch
    blk = _xe(gpu, gpu.DTYPE_BYTE, 1, 80, 16, 32)  # (more than 64 rows)
    x = torch.zeros(2 * 32 * 80 * 16 * 2, dtype=torch.int8, device="cuda")
    out = torch.zeros(2 * blk.get_output_buffer_size(), 2, device="cuda")
    with pytest.raises(gpu.Mi355Error):
        blk.xcorrelate_n_device(2, x, out, stations_per_group=4)


def test_in_launch_reduction_two_streams_share_the_device(gpu, oracle):
    """Two fused launches of 256 workgroups each on two streams at once: only one workgroup fits a CU, so the units of a slice are no longer
    all resident together and the in-launch reduction's bounded waits and hand-overs do real work (the case a sharded pipeline's exchange
    or any other block's kernels create).  Every result of every launch bit-exact."""
    import torch
    N, F, T = 64, 1024, 256
    rng = np.random.default_rng(77)
    xs = [rng.integers(-128, 128, size=(T, N, F, 1, 2), dtype=np.int64).astype(np.int8) for _ in range(2)]
    refs = [oracle.xengine_ichar(N, F, 1, T, x.reshape(-1), exact=True) for x in xs]
    blks = [_xe(gpu, gpu.DTYPE_BYTE, 1, N, F, T) for _ in range(2)]
    dx = [torch.from_numpy(x).cuda() for x in xs]
    per = blks[0].get_output_buffer_size()
    rounds = 12
    outs = [[torch.zeros(per, 2, device="cuda") for _ in range(rounds)] for _ in range(2)]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    torch.cuda.synchronize()
    for r in range(rounds):
        for k in range(2):
            with torch.cuda.stream(streams[k]):
                blks[k].xcorrelate_device(dx[k], outs[k][r])
    torch.cuda.synchronize()
    for k in range(2):
        for r in range(rounds):
            assert np.array_equal(outs[k][r].cpu().numpy().view(np.complex64).reshape(-1), refs[k]), (k, r)


@pytest.mark.parametrize("N,F,T,npol,nint,W,shift", [(64, 1024, 512, 1, 1, 1, 0), (50, 1024, 512, 1, 1, 1, 5), (64, 512, 256, 1, 1, 1, 9), (32, 1024, 256, 2, 1, 1, 3),
                                                    (64, 2048, 128, 1, 1, 1, 14), (64, 1024, 128, 1, 3, 1, 2), (64, 1024, 128, 1, 2, 8, 11), (20, 512, 160, 1, 6, 1, 7)])
def test_early_touches_of_the_slow_lines(gpu, oracle, N, F, T, npol, nint, W, shift):
    """The fused kernel's early touches of the rows' slow 128-byte lines (prefetch_slow: rows of 8 / 16 / 32 whole lines, whole K blocks, at
    least four K blocks per time range): which lines and which lanes depends on the input's address bits 7..10, so the input is placed at
    several 128-byte shifts; one and several windows, reference and group-major layout, antennas that do not fill the row tiles.  Bit exact
    against the oracle with the touches on and off."""
    import torch
    rng = np.random.default_rng(N + F + T + shift)
    wins = rng.integers(-128, 128, size=(nint, T, N, F, npol, 2), dtype=np.int64).astype(np.int8)
    blk = _xe(gpu, gpu.DTYPE_BYTE, npol, N, F, T)
    per = blk.get_output_buffer_size()
    ref = np.concatenate([oracle.xengine_ichar(N, F, npol, T, wins[i].reshape(-1), exact=True) for i in range(nint)])
    Ng = N // W if W > 1 else 0
    host = np.ascontiguousarray(wins.reshape(nint, T, W, Ng, F, npol, 2).transpose(2, 0, 1, 3, 4, 5, 6)).reshape(-1) if W > 1 else wins.reshape(-1)
    raw = torch.zeros(host.size + 4096, dtype=torch.int8, device="cuda")
    x = raw[128 * shift:128 * shift + host.size]
    x.copy_(torch.from_numpy(host))
    out = torch.zeros(nint * per, 2, device="cuda")
    for off in (None, "1"):
        if off: os.environ["MI355_XE_NO_PREFETCH"] = off
        try:
            out.zero_()
            if nint > 1 or W > 1: blk.xcorrelate_n_device(nint, x, out, stations_per_group=Ng)
            else: blk.xcorrelate_device(x, out)
            torch.cuda.synchronize()
        finally:
            os.environ.pop("MI355_XE_NO_PREFETCH", None)
        assert np.array_equal(out.cpu().numpy().view(np.complex64).reshape(-1), ref), off


@pytest.mark.parametrize("route", ["k_xe_i8_lines<split>", "k_xe_i8_fused"])
def test_in_launch_reduction_recovers_from_failed_and_unfinished_launches(gpu, oracle, monkeypatch, route):
    """Both kernels that combine time ranges inside the launch (the whole-line kernel's tail, the default at these geometries, and the 32-byte-slice
    kernel's, MI355_XE_NO_LINES_SPLIT=1).  The arrival words of the in-launch reduction must not carry anything from one launch into the next but the launch count:
    (1) a call that fails before its kernel is enqueued (MI355_XE_FAIL_LAUNCH stands in for a bad stream handle / an exhausted device) returns
    an error and the next call on the handle is bit-exact; (2) a launch whose workgroups of some row lines leave before they arrive
    (MI355_XE_DBG bits 16 / 17: the other units wait, give up or finish alone) leaves counts that are short of full behind -- the
    following launches, which count in the other bank of words and clear this one, are bit-exact again.
    Single windows on the handle's workspace (config 5's kernel instance: ping-pong schedule + reduce-scatter tail) and batches of
    windows on the batch workspace."""
    import torch
    if route == "k_xe_i8_fused":
        monkeypatch.setenv("MI355_XE_NO_LINES_SPLIT", "1")
    N, F, T = 64, 1024, 512  # 64 slices x 4 time ranges = 256 workgroups, 4 K blocks per range
    rng = np.random.default_rng(91)
    blk = _xe(gpu, gpu.DTYPE_BYTE, 1, N, F, T)
    per = blk.get_output_buffer_size()
    xs = [rng.integers(-128, 128, size=T * N * F * 2, dtype=np.int64).astype(np.int8) for _ in range(2)]
    refs = [oracle.xengine_ichar(N, F, 1, T, x, exact=True) for x in xs]
    dx = [torch.from_numpy(x).cuda() for x in xs]
    count = [0]

    def one(expect_ok=True):
        k = count[0] & 1
        count[0] += 1
        out = torch.full((per, 2), 3.0, device="cuda")
        blk.xcorrelate_device(dx[k], out)
        torch.cuda.synchronize()
        if expect_ok:
            assert np.array_equal(out.cpu().numpy().view(np.complex64).reshape(-1), refs[k]), count[0]

    one(); one()
    r = blk.last_route()
    assert r["kernel"] == route and r["tsplit"] == 4 and r["in_launch_reduce"] == 1, r
    monkeypatch.setenv("MI355_XE_FAIL_LAUNCH", "1")
    with pytest.raises(gpu.Mi355Error):
        one()
    monkeypatch.delenv("MI355_XE_FAIL_LAUNCH")
    one(); one(); one()
    for bits in ("65536", "131072"):  # only / all but the workgroups of lines 3 and 11 run: every slice of the others is left unfinished
        monkeypatch.setenv("MI355_XE_DBG", bits)
        monkeypatch.setenv("MI355_XE_WAIT_US", "5")  # (nobody waits tens of microseconds for units that never come)
        one(expect_ok=False)
        monkeypatch.delenv("MI355_XE_DBG")
        monkeypatch.delenv("MI355_XE_WAIT_US")
        one(); one(); one()
    # the batch workspace: per-rank geometry, 8 windows x 8 slices x 4 ranges = 256 workgroups
    Fb, nint = 128, 8
    bb = _xe(gpu, gpu.DTYPE_BYTE, 1, N, Fb, T)
    perb = bb.get_output_buffer_size()
    w = rng.integers(-128, 128, size=(nint, T, N, Fb, 1, 2), dtype=np.int64).astype(np.int8)
    refb = np.concatenate([oracle.xengine_ichar(N, Fb, 1, T, w[i].reshape(-1), exact=True) for i in range(nint)])
    dw = torch.from_numpy(w).cuda()

    def batch(expect_ok=True):
        out = torch.full((nint * perb, 2), 3.0, device="cuda")
        bb.xcorrelate_n_device(nint, dw, out)
        torch.cuda.synchronize()
        if expect_ok:
            assert np.array_equal(out.cpu().numpy().view(np.complex64).reshape(-1), refb)

    batch(); batch()
    r = bb.last_route()
    assert r["kernel"] == route and r["tsplit"] == 4 and r["in_launch_reduce"] == 1, r
    monkeypatch.setenv("MI355_XE_FAIL_LAUNCH", "1")
    with pytest.raises(gpu.Mi355Error):
        batch()
    monkeypatch.delenv("MI355_XE_FAIL_LAUNCH")
    batch(); batch()
    monkeypatch.setenv("MI355_XE_DBG", "131072")
    monkeypatch.setenv("MI355_XE_WAIT_US", "5")
    batch(expect_ok=False)
    monkeypatch.delenv("MI355_XE_DBG")
    monkeypatch.delenv("MI355_XE_WAIT_US")
    batch(); batch(); batch()


def test_one_handle_on_two_streams_is_ordered_by_the_library(gpu, oracle):
    """One handle, calls alternating between two streams with no synchronisation by the caller: the launches share the handle's partial-sum
    workspace (inboxes and arrival words of the in-launch reduction), so the library orders them with an event whenever the stream changes
    (mi355_xengine::ws_stream).  Single windows and batches; every result bit-exact."""
    import torch
    N, F, T = 64, 1024, 512
    rng = np.random.default_rng(92)
    xs = [rng.integers(-128, 128, size=(T, N, F, 1, 2), dtype=np.int64).astype(np.int8) for _ in range(3)]
    refs = [oracle.xengine_ichar(N, F, 1, T, x.reshape(-1), exact=True) for x in xs]
    dx = [torch.from_numpy(x).cuda() for x in xs]
    blk = _xe(gpu, gpu.DTYPE_BYTE, 1, N, F, T)
    per = blk.get_output_buffer_size()
    rounds = 18
    outs = [torch.zeros(per, 2, device="cuda") for _ in range(rounds)]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    torch.cuda.synchronize()
    for r in range(rounds):
        with torch.cuda.stream(streams[r & 1]):
            blk.xcorrelate_device(dx[r % 3], outs[r])
    torch.cuda.synchronize()
    for r in range(rounds):
        assert np.array_equal(outs[r].cpu().numpy().view(np.complex64).reshape(-1), refs[r % 3]), r
    # batches of the per-rank geometry on the batch workspace, window counts alternating as well (the words move with the count)
    Fb, T = 128, 256
    bb = _xe(gpu, gpu.DTYPE_BYTE, 1, N, Fb, T)
    perb = bb.get_output_buffer_size()
    w = rng.integers(-128, 128, size=(8, T, N, Fb, 1, 2), dtype=np.int64).astype(np.int8)
    refb = np.concatenate([oracle.xengine_ichar(N, Fb, 1, T, w[i].reshape(-1), exact=True) for i in range(8)])
    dw = torch.from_numpy(w).cuda()
    outb = [torch.zeros(8 * perb, 2, device="cuda") for _ in range(10)]
    for r in range(10):
        n = 8 if r % 3 else 4
        with torch.cuda.stream(streams[r & 1]):
            bb.xcorrelate_n_device(n, dw, outb[r])
    torch.cuda.synchronize()
    for r in range(10):
        n = 8 if r % 3 else 4
        assert np.array_equal(outb[r].cpu().numpy().view(np.complex64).reshape(-1)[:n * perb], refb[:n * perb]), r


@pytest.mark.parametrize("N,F,T,npol,nint", [(64, 1024, 128, 1, 8), (64, 512, 128, 1, 12), (64, 512, 160, 1, 16), (64, 256, 128, 1, 48),
                                             (50, 1024, 128, 1, 8), (32, 1024, 128, 2, 8), (64, 1024, 128, 1, 5)])
def test_batched_persistent_workgroups(gpu, oracle, monkeypatch, N, F, T, npol, nint):
    """More whole-integration units than CUs: a workgroup runs its units one after the other (FuArgs::items) -- the same slice of windows
    w, w + grid / (4 lines), ... -- and requests a unit's first K blocks while it stores the previous unit's matrix.  Every window bit-exact
    against the oracle and identical to the one-unit-per-workgroup launch (MI355_XE_NO_PERSIST); accumulate; window counts that do not
    divide evenly fall back to one unit per workgroup."""
    import torch
    rng = np.random.default_rng(N * 13 + nint)
    wins = rng.integers(-128, 128, size=(nint, T, N, F, npol, 2), dtype=np.int64).astype(np.int8)
    blk = _xe(gpu, gpu.DTYPE_BYTE, npol, N, F, T)
    per = blk.get_output_buffer_size()
    ref = np.concatenate([oracle.xengine_ichar(N, F, npol, T, wins[i].reshape(-1), exact=True) for i in range(nint)])
    x = torch.from_numpy(wins).cuda()
    out = torch.zeros(nint * per, 2, device="cuda")
    blk.xcorrelate_n_device(nint, x, out)
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy().view(np.complex64).reshape(-1), ref)
    monkeypatch.setenv("MI355_XE_NO_PERSIST", "1")
    plain = torch.zeros(nint * per, 2, device="cuda")
    blk.xcorrelate_n_device(nint, x, plain)
    torch.cuda.synchronize()
    monkeypatch.delenv("MI355_XE_NO_PERSIST")
    assert torch.equal(out, plain)
    blk.xcorrelate_n_device(nint, x, out, accumulate=True)
    torch.cuda.synchronize()
    ref2 = np.concatenate([oracle.xengine_ichar(N, F, npol, T, wins[i].reshape(-1), exact=True, acc=ref[i * per:(i + 1) * per].copy()) for i in range(nint)])
    assert np.array_equal(out.cpu().numpy().view(np.complex64).reshape(-1), ref2)


def test_ichar_scale_single_precision_is_exact(gpu):
    """The matrix stores of the fused IChar kernels scale sums below 2^24 in single precision (q = fl(S c), r = S - 16129 q by one fma,
    fl(q + r c)); the reference's arithmetic is (float)((double)S / 127 / 127) (lib/clXEngine_impl.cc:859-867 applied to every product).
    The device compares the two for EVERY |S| <= 2^24: no sum may differ in a single bit."""
    blk = _xe(gpu, gpu.DTYPE_BYTE, 1, 4, 16, 32)
    assert blk.selftest_scale() == 0


_DEAD_STREAM_SCRIPT = r'''
import ctypes, os, sys
sys.path.insert(0, os.environ["MI355_REPO"])
import numpy as np, torch
import __graft_entry__ as e
pkg, o = e.load_package(), e.load_oracle()
path = [l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l][0]
hip = ctypes.CDLL(path)
N, F, T = 64, 128, 256   # four time ranges: the launches share the handle's partial-sum workspace
rng = np.random.default_rng(4)
x = rng.integers(-128, 128, size=(T, N, F, 1, 2), dtype=np.int64).astype(np.int8)
ref = o.xengine_ichar(N, F, 1, T, x.reshape(-1), exact=True)
blk = pkg.clXEngine(1, 2, 0, 0, False, pkg.DTYPE_BYTE, 1, N, 1, 0, F, T, [])
dx = torch.from_numpy(x).cuda()
outs = [torch.zeros(blk.get_output_buffer_size(), 2, device="cuda") for _ in range(6)]
torch.cuda.synchronize()
for r in range(3):
    s = ctypes.c_void_p()
    assert hip.hipStreamCreate(ctypes.byref(s)) == 0
    with torch.cuda.stream(torch.cuda.ExternalStream(s.value)):
        blk.xcorrelate_device(dx, outs[2 * r])
    assert hip.hipStreamDestroy(s) == 0      # (not synchronised: the launch may still be in flight)
    blk.xcorrelate_device(dx, outs[2 * r + 1])  # torch's current stream: another one than the workspace's last, which no longer exists
torch.cuda.synchronize()
for r, v in enumerate(outs):
    assert np.array_equal(v.cpu().numpy().view(np.complex64).reshape(-1), ref), r
print("dead-stream ok")
'''


def test_handle_survives_a_destroyed_stream(gpu):
    """The workspace's previous stream was destroyed before the next call on the handle (a per-call stream): the library must not fail (and must
    not stay broken), it waits for the device instead of the dead stream.  Run in a child process -- the test hands the runtime a dead handle."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", _DEAD_STREAM_SCRIPT], capture_output=True, text=True, timeout=300, env=dict(os.environ, MI355_REPO=root))
    assert r.returncode == 0 and "dead-stream ok" in r.stdout, (r.returncode, r.stdout[-500:], r.stderr[-1500:])
