"""GPU parity: clPolyphaseChannelizer through the C ABI vs the oracle and the float64
closed form.  The reference holds no vectors for this block (parity
unpinned by the reference): the anchors are an independent implementation
(scipy.signal.upfirdn, independent_golden.npz) and the closed-form golden fixtures."""
import numpy as np
import pytest

from conftest import GPU_ARGS, crandn, golden, relerr

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _run(gpu, taps, buf, M, R, chmap, xh):
    blk = gpu.clPolyphaseChannelizer(*GPU_ARGS, taps, buf, M, R, chmap)
    y = np.empty(blk.noutput(), np.complex64)
    assert blk.ninput() == buf - R + len(taps)
    assert blk.general_work(y.size, [xh.size], [xh], [y]) == len(chmap) * buf // R
    return y


def test_golden_closed_form_cases(gpu):
    g = golden("pfb_golden.npz")
    for c in "abc":  # a: reference flowgraph M=3,R=2,145 taps; b: config-4 shape M=64; c: oversampled M=8,R=4 permuted map
        M, R, buf = (int(v) for v in g[c + "_cfg"])
        y = _run(gpu, g[c + "_taps"], buf, M, R, g[c + "_chmap"], g[c + "_x"])
        assert relerr(y, g[c + "_y"]) <= TOL, c


def test_independent_scipy_cases(gpu):
    """The HIP path against data no author of it (or of the oracle) wrote the arithmetic for: scipy.signal.upfirdn, one channel at a
    time -- mix down, FIR, decimate (tests/golden/gen_golden.py::independent_golden).  pa: the reference flowgraph's 3 channels at
    R = 2; pb: BASELINE config 4's shape (64 channels x 32 taps per arm); pc / pd: 2-fold and 4-fold oversampled maps."""
    g = golden("independent_golden.npz")
    for c in ("pa", "pb", "pc", "pd"):
        M, R, buf = (int(v) for v in g[c + "_cfg"])
        y = _run(gpu, g[c + "_taps"], buf, M, R, g[c + "_chmap"], g[c + "_x"])
        assert relerr(y, g[c + "_y"]) <= TOL, c


@pytest.mark.parametrize("M", [2, 4, 8, 16, 32, 64, 128, 256, 512])  # (512: the ring kernel up to 32 taps per arm, two kernels above)
@pytest.mark.parametrize("per_arm", [1, 5, 8, 13, 32, 33, 64])
def test_fast_path_all_channel_counts(gpu, oracle, M, per_arm):
    rng = np.random.default_rng(M * 100 + per_arm)
    K = M * per_arm - (3 if per_arm > 1 and M > 4 else 0)  # ragged last arm
    taps = (rng.standard_normal(K) / np.sqrt(per_arm)).astype(np.float32)
    buf = M * (4096 // M + 37)  # more than one workgroup iteration, ragged
    xh = crandn(rng, buf - M + K)
    chmap = list(range(M))
    y = _run(gpu, taps, buf, M, M, chmap, xh)
    assert relerr(y, oracle.pfb(taps, buf, M, M, chmap, xh, f64=True)) <= TOL


def test_channel_map_gather_with_duplicates(gpu, oracle):
    rng = np.random.default_rng(11)
    M, K, buf = 64, 2048, 64 * 100
    taps = rng.standard_normal(K).astype(np.float32) / 8
    xh = crandn(rng, buf - M + K)
    for chmap in ([5], [63, 0, 1, 1, 32], list(range(63, -1, -1))):
        y = _run(gpu, taps, buf, M, M, chmap, xh)
        assert relerr(y, oracle.pfb(taps, buf, M, M, chmap, xh, f64=True)) <= TOL


@pytest.mark.parametrize("M,R", [(3, 2), (3, 3), (8, 4), (16, 8), (5, 5), (12, 3), (512, 512)])
def test_generic_path_oversampled_and_odd(gpu, oracle, M, R):
    rng = np.random.default_rng(M * 7 + R)
    K = 7 * M + 1
    taps = rng.standard_normal(K).astype(np.float32) / 3
    lcm = M * R // np.gcd(M, R)
    buf = lcm * 11
    xh = crandn(rng, buf - R + K)
    chmap = [M - 1, 0] + list(range(min(M, 3)))
    y = _run(gpu, taps, buf, M, R, chmap, xh)
    assert relerr(y, oracle.pfb(taps, buf, M, R, chmap, xh, f64=True)) <= TOL


# channel counts outside the specialised kernels: the branch filters register-tiled (critically sampled, 2- and 4-fold oversampled; any
# other ratio one output per thread), the M-point DFT as a clFFT transform (power of two above 256, mixed radix, chirp-z for a prime count)
# when 16 or more channels are mapped, M products per output below that; identity, scrambled and short channel maps
@pytest.mark.parametrize("M,R,per_arm,nmap", [(10, 10, 17, 10), (20, 20, 3, 20), (48, 48, 70, 48), (100, 100, 17, 100), (100, 50, 33, 100),
                                              (100, 25, 9, 37), (100, 30, 5, 100), (200, 200, 17, 16), (200, 200, 3, 15), (211, 211, 6, 211),
                                              (512, 512, 70, 512), (1000, 1000, 4, 999), (1024, 512, 9, 1024), (64, 32, 80, 64), (256, 256, 65, 256)])
def test_generic_path_large_and_unusual_channel_counts(gpu, oracle, monkeypatch, M, R, per_arm, nmap):
    rng = np.random.default_rng(M * 7 + R + per_arm)
    K = M * per_arm - (M // 3 if per_arm > 1 else 0)  # ragged last arm
    taps = (rng.standard_normal(K) / np.sqrt(per_arm)).astype(np.float32)
    steps = 83
    buf = steps * R
    while buf % M:  # (the reference asks for whole frames of M items per buffer, :59-62)
        steps += 1
        buf = steps * R
    xh = crandn(rng, buf - R + K)
    chmap = list(range(M)) if nmap == M else rng.permutation(M)[:nmap].tolist()
    ref = oracle.pfb(taps, buf, M, R, chmap, xh, f64=True)
    assert relerr(_run(gpu, taps, buf, M, R, chmap, xh), ref) <= TOL
    monkeypatch.setenv("MI355_PFB_BRANCHES_PER_OUTPUT", "1")  # the per-output branch kernel (read per call) ...
    monkeypatch.setenv("MI355_PFB_DIRECT_DFT", "1")            # ... and the direct DFT (read at create) on the same input
    assert relerr(_run(gpu, taps, buf, M, R, chmap, xh), ref) <= TOL


# critically sampled, a channel count with a one-pass mixed-radix transform, at most 32 taps per arm: branch filters and transform in ONE kernel
# (k_pfb_mr, fft_mr.hip) -- several iterations per time range, ranges that end beyond the call, last passes that store for themselves (runs of
# at least 64 bytes: 100, 360, 500 channels) and through LDS (20, 24, 48, 96, 30), whole and partial channel maps (fewer than 16 channels
# mapped too), fewer taps than channels, a call shorter than one
# iteration; against the oracle and against the two-kernel form of the same handle (MI355_PFB_NO_MR_FUSED, read per call: same operation
# order in the filters, the same passes in the transform -- the compiler contracts a few multiply-adds differently, hence not bit for bit)
@pytest.mark.parametrize("M,per_arm,nmap,steps", [(100, 32, 100, 700), (20, 8, 20, 3001), (24, 9, 16, 517), (48, 32, 48, 260), (360, 5, 360, 90),
                                                  (500, 32, 250, 41), (96, 16, 96, 1500), (30, 4, 30, 64), (100, 32, 100, 7), (12, 3, 12, 1024),
                                                  (504, 3, 504, 20), (6, 32, 6, 300), (14, 1, 14, 100), (10, 17, 3, 500)])
def test_filters_and_transform_in_one_kernel(gpu, oracle, monkeypatch, M, per_arm, nmap, steps):
    rng = np.random.default_rng(M * 11 + per_arm + steps)
    K = M * per_arm - (M // 3 if per_arm % 2 else 0)  # ragged last arm for the odd tap counts
    taps = (rng.standard_normal(K) / np.sqrt(per_arm)).astype(np.float32)
    buf = steps * M
    xh = crandn(rng, buf - M + K)
    chmap = list(range(M)) if nmap == M else rng.permutation(M)[:nmap].tolist()
    ref = oracle.pfb(taps, buf, M, M, chmap, xh, f64=True)
    y = _run(gpu, taps, buf, M, M, chmap, xh)
    assert relerr(y, ref) <= TOL
    monkeypatch.setenv("MI355_PFB_NO_MR_FUSED", "1")
    y2 = _run(gpu, taps, buf, M, M, chmap, xh)
    assert relerr(y2, ref) <= TOL
    assert relerr(y, y2) <= 1e-6


@pytest.mark.parametrize("M,per_arm,steps", [(100, 32, 333), (20, 8, 1001), (48, 16, 77), (360, 5, 19), (10, 32, 4099)])
def test_one_kernel_form_stays_inside_its_buffers(gpu, oracle, M, per_arm, steps):
    """Input and output of k_pfb_mr sit inside larger allocations: NaNs around the input (a read outside that reached an output would show), a sentinel
    around the output (ranges that end beyond the call, the unconditional part of the store loops)."""
    import torch
    rng = np.random.default_rng(M + steps)
    taps = (rng.standard_normal(M * per_arm) / np.sqrt(per_arm)).astype(np.float32)
    buf = steps * M
    blk = gpu.clPolyphaseChannelizer(*GPU_ARGS, taps, buf, M, M, list(range(M)))
    nin, nout, pad = blk.ninput(), blk.noutput(), 4096
    big = torch.full((nin + 2 * pad, 2), float("nan"), device="cuda")
    xh = crandn(rng, nin)
    big[pad:pad + nin] = torch.from_numpy(xh.view(np.float32).reshape(-1, 2)).cuda()
    ybig = torch.full((nout + 2 * pad, 2), 7.0, device="cuda")
    blk.work_device([big[pad:pad + nin]], [ybig[pad:pad + nout]])
    torch.cuda.synchronize()
    assert torch.all(ybig[:pad] == 7.0) and torch.all(ybig[pad + nout:] == 7.0)
    y = ybig[pad:pad + nout].cpu().numpy().view(np.complex64).reshape(-1)
    assert np.all(np.isfinite(y.view(np.float32)))
    assert relerr(y, oracle.pfb(taps, buf, M, M, list(range(M)), xh, f64=True)) <= TOL


# the two-kernel path (more than 512 channels, ...): the streaming branch-filter kernel k_pfb_fir (register ring rotated by phases, <= 8 and 17 ... 32
# taps per arm) against the register-tiled k_pfb_branches_t (MI355_PFB_NO_FIR_RING, read per call) -- the same operation order, so bit for bit -- and
# against the oracle; arm blocks with a ragged last block, ranges that end beyond the call, fewer arms than a workgroup has threads, a prime count
@pytest.mark.parametrize("M,per_arm,steps", [(1024, 32, 300), (1000, 7, 333), (600, 25, 100), (2048, 4, 50), (7, 5, 3000), (768, 17, 9), (1024, 1, 40)])
def test_streaming_branch_filters_equal_the_tiled_ones(gpu, oracle, monkeypatch, M, per_arm, steps):
    rng = np.random.default_rng(M + per_arm)
    K = M * per_arm - (M // 3 if per_arm % 2 and per_arm > 1 else 0)
    taps = (rng.standard_normal(K) / np.sqrt(per_arm)).astype(np.float32)
    buf = steps * M
    xh = crandn(rng, buf - M + K)
    chmap = list(range(M))
    y = _run(gpu, taps, buf, M, M, chmap, xh)
    assert relerr(y, oracle.pfb(taps, buf, M, M, chmap, xh, f64=True)) <= TOL
    monkeypatch.setenv("MI355_PFB_NO_FIR_RING", "1")
    assert np.array_equal(y, _run(gpu, taps, buf, M, M, chmap, xh))


# 2- / 4-fold oversampled channelizers with 64 / 128 / 256 channels and <= 32 taps per arm run on the ring kernel, one launch per residue of
# the step number (quarter-turn factors on the channels); identity and scrambled maps; step counts that leave the residues uneven; and the
# generic path on the same input (MI355_PFB_NO_FAST_OVERSAMPLED)
@pytest.mark.parametrize("M,R,per_arm,nmap,steps", [(64, 32, 8, 64, 203), (64, 32, 32, 64, 64), (64, 16, 13, 64, 205), (128, 64, 16, 128, 99),
                                                    (128, 32, 32, 50, 134), (256, 128, 8, 256, 77), (256, 64, 20, 256, 42), (64, 32, 5, 7, 1001)])
def test_oversampled_ring_kernel(gpu, oracle, monkeypatch, M, R, per_arm, nmap, steps):
    rng = np.random.default_rng(M + R + per_arm + steps)
    K = M * per_arm - (M // 5 if per_arm > 1 else 0)
    taps = (rng.standard_normal(K) / np.sqrt(per_arm)).astype(np.float32)
    buf = steps * R
    while buf % M:
        steps += 1
        buf = steps * R
    xh = crandn(rng, buf - R + K)
    chmap = list(range(M)) if nmap == M else rng.permutation(M)[:nmap].tolist()
    ref = oracle.pfb(taps, buf, M, R, chmap, xh, f64=True)
    assert relerr(_run(gpu, taps, buf, M, R, chmap, xh), ref) <= TOL
    monkeypatch.setenv("MI355_PFB_NO_FAST_OVERSAMPLED", "1")  # read at create
    assert relerr(_run(gpu, taps, buf, M, R, chmap, xh), ref) <= TOL


def test_baseline_config4_shape_and_streaming(gpu, oracle):
    """64 channels x 32 taps/arm, buf_items 65536 (BASELINE configs[3]); two consecutive calls
    with GNU Radio's history equal one double-length call."""
    taps = np.concatenate([oracle.firdes_low_pass(1.0, 64.0, 0.5, 0.0753), [0.0]]).astype(np.float32)
    assert taps.size == 2048
    rng = np.random.default_rng(1000)
    M, buf = 64, 65536
    x = crandn(rng, 2 * buf + 2048 - M)
    chmap = list(range(M))
    one = gpu.clPolyphaseChannelizer(*GPU_ARGS, taps, buf, M, M, chmap)
    two = gpu.clPolyphaseChannelizer(*GPU_ARGS, taps, 2 * buf, M, M, chmap)
    ya, yb, yw = np.empty(buf, np.complex64), np.empty(buf, np.complex64), np.empty(2 * buf, np.complex64)
    one.general_work(buf, [0], [x[:one.ninput()]], [ya])
    one.general_work(buf, [0], [x[buf:buf + one.ninput()]], [yb])  # scheduler consumed buf_items (:105)
    two.general_work(2 * buf, [0], [x], [yw])
    assert np.array_equal(np.concatenate([ya, yb]), yw)
    ref = oracle.pfb(taps, 64 * 40, M, M, chmap, x, f64=True)
    assert relerr(yw[:ref.size], ref) <= TOL
    # a tone in channel 5 comes out of channel 5 (channel 0 = centre, increasing c = increasing frequency)
    n = np.arange(x.size)
    tone = np.exp(2j * np.pi * 5 / 64 * n).astype(np.complex64)
    two.general_work(2 * buf, [0], [tone], [yw])
    p = (np.abs(yw.reshape(-1, 64)[64:]) ** 2).mean(0)
    assert p.argmax() == 5 and p[5] > 1e3 * np.delete(p, 5).max()


@pytest.mark.parametrize("M,tpa,ident", [(64, 32, True), (64, 8, False), (128, 16, True), (256, 8, True), (256, 32, True), (32, 16, True), (12, 5, False),
                                         (100, 32, True), (20, 8, True), (48, 16, True), (360, 3, True),  # (these four: k_pfb_mr, k buffers = one stream)
                                         (512, 32, True), (512, 8, False)])
def test_batched_call_equals_single_calls(gpu, oracle, M, tpa, ident, monkeypatch):
    """work_device(nbuf=k) -- general_work() offered k output multiples -- returns exactly the samples of k single calls, and the
    small-call schedule (k_pfbq: one workgroup per 16-step group) exactly those of the ring kernel (MI355_PFB_SMALL=0)."""
    import torch
    rng = np.random.default_rng(M * 100 + tpa)
    taps = rng.standard_normal(M * tpa).astype(np.float32)
    chmap = list(range(M)) if ident else [int(c) for c in rng.permutation(M)[:max(1, M // 2)]]
    buf, k = M * 48, 5
    blk = gpu.clPolyphaseChannelizer(*GPU_ARGS, taps, buf, M, M, chmap)
    x = torch.randn((k * buf + taps.size - M), 2, device="cuda")
    ys = torch.empty(k * blk.noutput(), 2, device="cuda")
    yb = torch.empty_like(ys)
    for b in range(k):
        blk.work_device([x[b * buf:]], [ys[b * blk.noutput():]])
    assert blk.work_device([x], [yb], nbuf=k) == k * blk.noutput()
    torch.cuda.synchronize()
    assert torch.equal(ys, yb)
    ref = oracle.pfb(taps, buf, M, M, chmap, x[:blk.ninput()].cpu().numpy().view(np.complex64).reshape(-1), f64=True)
    assert relerr(yb[:ref.size].cpu().numpy().view(np.complex64).reshape(-1), ref) <= TOL
    monkeypatch.setenv("MI355_PFB_SMALL", "0")
    ring = gpu.clPolyphaseChannelizer(*GPU_ARGS, taps, buf, M, M, chmap)
    yr = torch.empty_like(ys)
    ring.work_device([x], [yr], nbuf=k)
    torch.cuda.synchronize()
    assert torch.equal(yr, yb)


@pytest.mark.parametrize("M,R,tpa,ident", [(64, 32, 32, True), (128, 32, 16, False), (256, 128, 8, True)])
def test_batched_oversampled_equals_single_calls(gpu, oracle, M, R, tpa, ident):
    """The same for 2- / 4-fold oversampled channelizers on the ring kernel (one launch per residue of the step number and buffer): k
    buffers in one call == k single calls bit for bit, and every buffer matches the oracle on its span of the stream."""
    import torch
    rng = np.random.default_rng(M + R + tpa)
    taps = rng.standard_normal(M * tpa).astype(np.float32)
    chmap = list(range(M)) if ident else [int(c) for c in rng.permutation(M)[:M // 2]]
    buf, k = M * 24, 4
    blk = gpu.clPolyphaseChannelizer(*GPU_ARGS, taps, buf, M, R, chmap)
    x = torch.randn((k * buf + taps.size - R), 2, device="cuda")
    ys = torch.empty(k * blk.noutput(), 2, device="cuda")
    yb = torch.empty_like(ys)
    for b in range(k):
        blk.work_device([x[b * buf:]], [ys[b * blk.noutput():]])
    assert blk.work_device([x], [yb], nbuf=k) == k * blk.noutput()
    torch.cuda.synchronize()
    assert torch.equal(ys, yb)
    xh = x.cpu().numpy().view(np.complex64).reshape(-1)
    for b in (0, k - 1):
        ref = oracle.pfb(taps, buf, M, R, chmap, xh[b * buf:b * buf + blk.ninput()], f64=True)
        assert relerr(yb[b * blk.noutput():(b + 1) * blk.noutput()].cpu().numpy().view(np.complex64).reshape(-1), ref) <= TOL


def test_device_path_full_size_linearity(gpu, oracle):
    import torch
    taps = np.concatenate([oracle.firdes_low_pass(1.0, 64.0, 0.5, 0.0753), [0.0]]).astype(np.float32)
    M, buf = 64, 1 << 24
    blk = gpu.clPolyphaseChannelizer(*GPU_ARGS, taps, buf, M, M, list(range(M)))
    g = torch.Generator(device="cuda").manual_seed(4)
    a = torch.randn(blk.ninput(), 2, device="cuda", generator=g)
    b = torch.randn(blk.ninput(), 2, device="cuda", generator=g)
    ya, yb, yab = (torch.empty(blk.noutput(), 2, device="cuda") for _ in range(3))
    blk.work_device([a], [ya]); blk.work_device([b], [yb]); blk.work_device([2 * a - 3 * b], [yab])
    torch.cuda.synchronize()
    assert torch.allclose(yab, 2 * ya - 3 * yb, atol=1e-4, rtol=1e-4)
    xs = a[:2048 - 64 + 64 * 50].cpu().numpy().view(np.complex64).reshape(-1)
    ref = oracle.pfb(taps, 64 * 50, M, M, list(range(M)), xs, f64=True)
    assert relerr(ya[:ref.size].cpu().numpy().view(np.complex64).reshape(-1), ref) <= TOL


def test_constructor_errors(gpu):
    with pytest.raises(ValueError):
        gpu.clPolyphaseChannelizer(*GPU_ARGS, [1.0] * 8, 10, 4, 4, [0])  # lib/clPolyphaseChannelizer_impl.cc:59-62
    with pytest.raises(gpu.Mi355Error):
        gpu.clPolyphaseChannelizer(*GPU_ARGS, [1.0] * 8, 16, 4, 4, [4])  # channel outside 0..M-1
