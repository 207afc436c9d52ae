"""GPU parity: the whole-line form of the fused IChar X-engine (csrc/xengine_lines.hip: 64 stations, one polarisation, whole 128-byte lines per
request, the ten row-tile pairs split over four workgroups) against the oracle's exact mode, bit for bit, and against the 32-byte-slice kernel
(MI355_XE_NO_LINES=1).  Reference behaviour: lib/clXEngine_impl.cc:708-817, :859-867.  The path is chosen by mi355_xe_lines_ok (enough units to
fill the device without time ranges); MI355_XE_LINES_MIN_UNITS lowers that bar so that small geometries reach the kernel too."""
import os
import numpy as np
import pytest

from conftest import GPU_ARGS

pytestmark = pytest.mark.gpu


def _xe(gpu, N, F, T):
    return gpu.clXEngine(*GPU_ARGS, False, gpu.DTYPE_BYTE, 1, N, gpu.CLXCORR_TRIANGULAR_ORDER, 0, F, T, [])


def _run(gpu, blk, nint, x, out, ng=0):
    import torch
    if ng: blk.xcorrelate_n_device(nint, x, out, stations_per_group=ng)
    else: blk.xcorrelate_n_device(nint, x, out)
    torch.cuda.synchronize()


@pytest.fixture
def small_units(monkeypatch):
    monkeypatch.setenv("MI355_XE_LINES_MIN_UNITS", "4")


@pytest.mark.parametrize("F,T,nint", [(64, 32, 1), (64, 64, 2), (128, 96, 3), (192, 160, 1), (256, 32, 8), (512, 64, 16), (1024, 32, 5)])
def test_whole_line_kernel_bit_exact(gpu, oracle, small_units, F, T, nint):
    """Every window bit exact against the oracle: one and several K blocks, one unit and several units per workgroup (512 x 16: two), unit counts
    that are and are not a multiple of 32 (the pinned and the plain workgroup map)."""
    import torch
    N = 64
    rng = np.random.default_rng(F + 7 * T + nint)
    wins = rng.integers(-128, 128, size=(nint, T, N, F, 1, 2), dtype=np.int64).astype(np.int8)
    blk = _xe(gpu, N, F, T)
    per = blk.get_output_buffer_size()
    ref = np.concatenate([oracle.xengine_ichar(N, F, 1, T, wins[i].reshape(-1), exact=True) for i in range(nint)])
    x = torch.from_numpy(wins).cuda()
    out = torch.zeros(nint * per, 2, device="cuda")
    _run(gpu, blk, nint, x, out)
    got = out.cpu().numpy().view(np.complex64).reshape(-1)
    assert np.array_equal(got, ref)
    # the 32-byte-slice kernel gives the same bits
    os.environ["MI355_XE_NO_LINES"] = "1"
    try:
        old = torch.zeros_like(out)
        _run(gpu, blk, nint, x, old)
    finally:
        os.environ.pop("MI355_XE_NO_LINES", None)
    assert torch.equal(out, old)


@pytest.mark.parametrize("W", [2, 4, 8])
def test_whole_line_kernel_group_major(gpu, oracle, small_units, W):
    """The input as one all-to-all delivers it, [group][window][t][station in group][chan]: read in place (groups of 32, 16 and 8 stations)."""
    import torch
    N, F, T, nint = 64, 128, 64, 3
    rng = np.random.default_rng(W)
    wins = rng.integers(-128, 128, size=(nint, T, N, F, 1, 2), dtype=np.int64).astype(np.int8)
    blk = _xe(gpu, N, F, T)
    per = blk.get_output_buffer_size()
    ref = np.concatenate([oracle.xengine_ichar(N, F, 1, T, wins[i].reshape(-1), exact=True) for i in range(nint)])
    Ng = N // W
    x = torch.from_numpy(np.ascontiguousarray(wins.reshape(nint, T, W, Ng, F, 1, 2).transpose(2, 0, 1, 3, 4, 5, 6))).cuda()
    out = torch.zeros(nint * per, 2, device="cuda")
    _run(gpu, blk, nint, x, out, ng=Ng)
    assert np.array_equal(out.cpu().numpy().view(np.complex64).reshape(-1), ref)


def test_whole_line_kernel_extremes(gpu, oracle, small_units):
    """All samples -128 (the value whose negation does not exist in int8), all +127, and alternating signs, over the longest integration the
    kernel takes (16384 frames: the combined accumulator of a diagonal tile pair holds re + im, |.| <= T * 2^16)."""
    import torch
    N, F, T = 64, 64, 16384
    blk = _xe(gpu, N, F, T)
    per = blk.get_output_buffer_size()
    for fill in ("min", "max", "alt"):
        w = np.empty((T, N, F, 1, 2), np.int8)
        if fill == "min": w[:] = -128
        elif fill == "max": w[:] = 127
        else:
            w[..., 0] = -128
            w[..., 1] = 127
            w[::2, 1::2] = np.array([127, -128], np.int8)
        ref = oracle.xengine_ichar(N, F, 1, T, w.reshape(-1), exact=True)
        x = torch.from_numpy(w).cuda()
        out = torch.zeros(per, 2, device="cuda")
        _run(gpu, blk, 1, x, out)
        assert np.array_equal(out.cpu().numpy().view(np.complex64).reshape(-1), ref), fill


def test_whole_line_kernel_refusals_fall_back(gpu, oracle, small_units):
    """Geometries outside the kernel's range still give the oracle's bits through the 32-byte-slice kernel: 60 stations, 96 channels (rows that are
    not whole lines), a ragged integration, accumulate."""
    import torch
    for N, F, T in ((60, 64, 64), (64, 96, 64), (64, 64, 40)):
        rng = np.random.default_rng(N + F + T)
        w = rng.integers(-128, 128, size=(2, T, N, F, 1, 2), dtype=np.int64).astype(np.int8)
        blk = _xe(gpu, N, F, T)
        per = blk.get_output_buffer_size()
        ref = np.concatenate([oracle.xengine_ichar(N, F, 1, T, w[i].reshape(-1), exact=True) for i in range(2)])
        out = torch.zeros(2 * per, 2, device="cuda")
        _run(gpu, blk, 2, torch.from_numpy(w).cuda(), out)
        assert np.array_equal(out.cpu().numpy().view(np.complex64).reshape(-1), ref)
    N, F, T = 64, 64, 64
    rng = np.random.default_rng(5)
    w = rng.integers(-128, 128, size=(1, T, N, F, 1, 2), dtype=np.int64).astype(np.int8)
    blk = _xe(gpu, N, F, T)
    per = blk.get_output_buffer_size()
    x = torch.from_numpy(w).cuda()
    out = torch.zeros(per, 2, device="cuda")
    _run(gpu, blk, 1, x, out)
    ref = oracle.xengine_ichar(N, F, 1, T, w.reshape(-1), exact=True)
    blk.xcorrelate_n_device(1, x, out, accumulate=True)
    torch.cuda.synchronize()
    ref2 = oracle.xengine_ichar(N, F, 1, T, w.reshape(-1), exact=True, acc=ref.copy())
    assert np.array_equal(out.cpu().numpy().view(np.complex64).reshape(-1), ref2)


@pytest.mark.parametrize("nint,pub", [(8, "1"), (24, "1"), (12, "0"), (64, "1")])
def test_whole_line_kernel_default_route_many_units(gpu, oracle, monkeypatch, nint, pub):
    """The default route (no test switch) at 1024 channels x 64 frames: 2, 6, 3 and 16 units per workgroup, paced through the progress words (kept in
    the XCD's L2, or -- MI355_XE_LINES_PUB=0 -- published at agent scope), a workgroup's k-th unit k lines further on: bit exact against the oracle
    and, window for window, against the 32-byte-slice kernel."""
    import torch
    monkeypatch.setenv("MI355_XE_LINES_PUB", pub)
    N, F, T = 64, 1024, 64
    rng = np.random.default_rng(nint)
    wins = rng.integers(-128, 128, size=(nint, T, N, F, 1, 2), dtype=np.int64).astype(np.int8)
    blk = _xe(gpu, N, F, T)
    per = blk.get_output_buffer_size()
    x = torch.from_numpy(wins).cuda()
    out = torch.zeros(nint * per, 2, device="cuda")
    _run(gpu, blk, nint, x, out)
    got = out.cpu().numpy().view(np.complex64).reshape(nint, -1)
    for i in (range(nint) if nint <= 12 else sorted(set(range(0, nint, 5)) | {nint - 1})):  # (the rest: through the other kernel, below)
        assert np.array_equal(got[i], oracle.xengine_ichar(N, F, 1, T, wins[i].reshape(-1), exact=True)), i
    os.environ["MI355_XE_NO_LINES"] = "1"
    try:
        old = torch.zeros_like(out)
        _run(gpu, blk, nint, x, old)
    finally:
        os.environ.pop("MI355_XE_NO_LINES", None)
    assert torch.equal(out, old)


@pytest.mark.parametrize("F,T,nint,pace", [(512, 256, 8, None), (2048, 192, 2, None), (1024, 320, 4, None), (1024, 320, 8, "2")])
def test_whole_line_kernel_early_touches(gpu, oracle, monkeypatch, F, T, nint, pace):
    """The default route with the early touches of the slow lines on (rows of 8, 32 and 16 lines, more K blocks than the touch distance; unpaced --
    the default with touches -- and paced): requests for data nobody reads must change nothing -- the first and the last window bit exact against the
    oracle, all of them identical with the touches off."""
    import torch
    if pace is not None: monkeypatch.setenv("MI355_XE_LINES_PACE", pace)
    N = 64
    g = torch.Generator(device="cuda").manual_seed(F + T)
    x = torch.randint(-128, 128, (nint, T, N, F, 1, 2), dtype=torch.int8, device="cuda", generator=g)
    blk = _xe(gpu, N, F, T)
    per = blk.get_output_buffer_size()
    out = torch.zeros(nint * per, 2, device="cuda")
    _run(gpu, blk, nint, x, out)
    os.environ["MI355_XE_LINES_PF"] = "0"
    try:
        off = torch.zeros_like(out)
        _run(gpu, blk, nint, x, off)
    finally:
        os.environ.pop("MI355_XE_LINES_PF", None)
    assert torch.equal(out, off)
    got = out.cpu().numpy().view(np.complex64).reshape(nint, -1)
    for i in (0, nint - 1):
        assert np.array_equal(got[i], oracle.xengine_ichar(N, F, 1, T, x[i].cpu().numpy().reshape(-1), exact=True)), i


@pytest.mark.parametrize("F,T,nint", [(1024, 64, 5), (1024, 64, 11), (512, 96, 13), (1024, 32, 3), (512, 64, 3)])
def test_window_counts_between_the_good_ones_are_split(gpu, oracle, monkeypatch, F, T, nint):
    """mi355_xengine_xcorrelate_n_dev cuts a call whose window count the whole-line kernel does not take (5 = 4 + 1, 11 = 8 + 2 + 1, 13 = 8 + 5, 3 = 2 + 1 where a window is a quarter of the device) into
    stream-ordered launches: every window bit exact, and identical to the one-launch form (MI355_XE_NO_SPLIT=1)."""
    import torch
    N = 64
    rng = np.random.default_rng(100 + nint)
    wins = rng.integers(-128, 128, size=(nint, T, N, F, 1, 2), dtype=np.int64).astype(np.int8)
    blk = _xe(gpu, N, F, T)
    per = blk.get_output_buffer_size()
    x = torch.from_numpy(wins).cuda()
    out = torch.zeros(nint * per, 2, device="cuda")
    _run(gpu, blk, nint, x, out)
    got = out.cpu().numpy().view(np.complex64).reshape(nint, -1)
    for i in range(nint):
        assert np.array_equal(got[i], oracle.xengine_ichar(N, F, 1, T, wins[i].reshape(-1), exact=True)), i
    monkeypatch.setenv("MI355_XE_NO_SPLIT", "1")
    one = torch.zeros_like(out)
    _run(gpu, blk, nint, x, one)
    assert torch.equal(out, one)
    # accumulate: the whole-line kernel does not take it, so the call is one launch of the 32-byte-slice kernel as before
    monkeypatch.delenv("MI355_XE_NO_SPLIT")
    blk.xcorrelate_n_device(nint, x, out, accumulate=True)
    torch.cuda.synchronize()
    got2 = out.cpu().numpy().view(np.complex64).reshape(nint, -1)
    ref2 = oracle.xengine_ichar(N, F, 1, T, wins[0].reshape(-1), exact=True, acc=got[0].copy())
    assert np.array_equal(got2[0], ref2)


def test_whole_line_kernel_config5_batch(gpu, oracle):
    """BASELINE config 5 (64 x 1024 x 1024), eight windows per launch -- the default route, two units per workgroup: the first and the last window bit
    exact against the oracle, every window identical to the 32-byte-slice kernel's."""
    import torch
    N, F, T, nint = 64, 1024, 1024, 8
    g = torch.Generator(device="cuda").manual_seed(3)
    x = torch.randint(-128, 128, (nint, T, N, F, 1, 2), dtype=torch.int8, device="cuda", generator=g)
    blk = _xe(gpu, N, F, T)
    per = blk.get_output_buffer_size()
    out = torch.zeros(nint * per, 2, device="cuda")
    _run(gpu, blk, nint, x, out)
    # the route this call must take (a wrong one is 1.4 x slower and was once only visible in bench.py): the whole-line kernel, one launch, two units per
    # workgroup on 256 workgroups, early touches of the slow lines on, the four workgroups of a line paced by two half K blocks
    r = blk.last_route()
    assert r["kernel"] == "k_xe_i8_lines" and r["launches"] == 1 and r["windows"] == nint and r["workgroups"] == 256 and r["units_per_workgroup"] == 2, r
    assert r["touches"] > 0 and r["pace"] == 2 and r["tsplit"] == 1, r
    os.environ["MI355_XE_NO_LINES"] = "1"
    try:
        old = torch.zeros_like(out)
        _run(gpu, blk, nint, x, old)
        r = blk.last_route()
        assert r["kernel"] == "k_xe_i8_fused" and r["windows"] == nint, r
    finally:
        os.environ.pop("MI355_XE_NO_LINES", None)
    assert torch.equal(out, old)
    got = out.cpu().numpy().view(np.complex64).reshape(nint, -1)
    for i in (0, nint - 1):
        ref = oracle.xengine_ichar(N, F, 1, T, x[i].cpu().numpy().reshape(-1), exact=True)
        assert np.array_equal(got[i], ref), i


# ---- time ranges (k_xe_i8_lines<true>): fewer (window, line, pair group) units than compute units -- the reference's one-integration-per-call shape
@pytest.fixture
def split_any(monkeypatch):
    monkeypatch.setenv("MI355_XE_LINES_SPLIT_ANY", "1")


def _route(gpu, blk):
    return blk.last_route()


@pytest.mark.parametrize("F,T,nint,S", [(64, 128, 1, 4), (64, 64, 1, 2), (128, 256, 2, 4), (192, 192, 1, 2), (256, 512, 1, 4), (512, 128, 3, 2)])
def test_time_ranges_bit_exact(gpu, oracle, split_any, monkeypatch, F, T, nint, S):
    """Two and four time ranges per team, combined inside the launch: bit exact against the oracle and against the 32-byte-slice kernel; unit counts
    that are and are not multiples of 32; a second launch on the same workspace (the other bank of arrival words)."""
    import torch
    monkeypatch.setenv("MI355_XE_TSPLIT", str(S))
    N = 64
    rng = np.random.default_rng(F + 3 * T + nint + S)
    wins = rng.integers(-128, 128, size=(nint, T, N, F, 1, 2), dtype=np.int64).astype(np.int8)
    blk = _xe(gpu, N, F, T)
    per = blk.get_output_buffer_size()
    ref = np.concatenate([oracle.xengine_ichar(N, F, 1, T, wins[i].reshape(-1), exact=True) for i in range(nint)])
    x = torch.from_numpy(wins).cuda()
    for rep in range(3):
        out = torch.zeros(nint * per, 2, device="cuda")
        _run(gpu, blk, nint, x, out)
        assert np.array_equal(out.cpu().numpy().view(np.complex64).reshape(-1), ref), rep
    r = _route(gpu, blk)
    if r is not None:
        assert r["kernel"] == "k_xe_i8_lines<split>" and r["tsplit"] == S, r
    if nint == 1:  # the one-window entry point takes the same route
        out = torch.zeros(per, 2, device="cuda")
        blk.xcorrelate_device(x[0], out)
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy().view(np.complex64).reshape(-1), ref)


@pytest.mark.parametrize("S", [2, 4])
def test_time_ranges_fallback_when_waits_run_out(gpu, oracle, split_any, monkeypatch, S):
    """MI355_XE_DBG=512: every bounded wait runs out at once, so every workgroup but the last of its team hands its own pieces over and leaves, and
    the last one finishes the whole team -- the path a team takes when its workgroups are not resident together."""
    import torch
    monkeypatch.setenv("MI355_XE_TSPLIT", str(S))
    monkeypatch.setenv("MI355_XE_DBG", "512")
    N, F, T = 64, 128, 256
    rng = np.random.default_rng(S)
    w = rng.integers(-128, 128, size=(1, T, N, F, 1, 2), dtype=np.int64).astype(np.int8)
    blk = _xe(gpu, N, F, T)
    ref = oracle.xengine_ichar(N, F, 1, T, w.reshape(-1), exact=True)
    x = torch.from_numpy(w).cuda()
    for rep in range(2):
        out = torch.zeros(blk.get_output_buffer_size(), 2, device="cuda")
        _run(gpu, blk, 1, x, out)
        assert np.array_equal(out.cpu().numpy().view(np.complex64).reshape(-1), ref), rep


def test_time_ranges_group_major_and_extremes(gpu, oracle, split_any, monkeypatch):
    """Group-major input (an all-to-all's receive buffer) through the time-range form; all -128 over 16384 frames in four ranges."""
    import torch
    monkeypatch.setenv("MI355_XE_TSPLIT", "4")
    N, F, T, nint, W = 64, 128, 128, 2, 4
    rng = np.random.default_rng(99)
    wins = rng.integers(-128, 128, size=(nint, T, N, F, 1, 2), dtype=np.int64).astype(np.int8)
    blk = _xe(gpu, N, F, T)
    per = blk.get_output_buffer_size()
    ref = np.concatenate([oracle.xengine_ichar(N, F, 1, T, wins[i].reshape(-1), exact=True) for i in range(nint)])
    Ng = N // W
    x = torch.from_numpy(np.ascontiguousarray(wins.reshape(nint, T, W, Ng, F, 1, 2).transpose(2, 0, 1, 3, 4, 5, 6))).cuda()
    out = torch.zeros(nint * per, 2, device="cuda")
    _run(gpu, blk, nint, x, out, ng=Ng)
    assert np.array_equal(out.cpu().numpy().view(np.complex64).reshape(-1), ref)
    N, F, T = 64, 64, 16384
    blk = _xe(gpu, N, F, T)
    w = np.full((T, N, F, 1, 2), -128, np.int8)
    ref = oracle.xengine_ichar(N, F, 1, T, w.reshape(-1), exact=True)
    out = torch.zeros(blk.get_output_buffer_size(), 2, device="cuda")
    _run(gpu, blk, 1, torch.from_numpy(w).cuda(), out)
    assert np.array_equal(out.cpu().numpy().view(np.complex64).reshape(-1), ref)


def test_config5_one_and_two_windows_take_the_time_range_form(gpu):
    """BASELINE config 5 as the reference's operator is called -- ONE integration per call (lib/clXEngine_impl.h:184-201) -- and two windows per call:
    the whole-line kernel with four / two time ranges, identical to the 32-byte-slice kernel's bits (which test_xengine_gpu.py pins on the oracle)."""
    import torch
    N, F, T = 64, 1024, 1024
    g = torch.Generator(device="cuda").manual_seed(5)
    blk = _xe(gpu, N, F, T)
    per = blk.get_output_buffer_size()
    for nint in (1, 2):
        x = torch.randint(-128, 128, (nint, T, N, F, 1, 2), dtype=torch.int8, device="cuda", generator=g)
        out = torch.zeros(nint * per, 2, device="cuda")
        for rep in range(3):
            out.zero_()
            if nint == 1: blk.xcorrelate_device(x, out)
            else: blk.xcorrelate_n_device(nint, x, out)
            torch.cuda.synchronize()
            r = _route(gpu, blk)
            if r is not None:
                assert r["kernel"] == "k_xe_i8_lines<split>" and r["tsplit"] == 4 // nint, r
            os.environ["MI355_XE_NO_LINES"] = "1"
            try:
                old = torch.zeros_like(out)
                if nint == 1: blk.xcorrelate_device(x, old)
                else: blk.xcorrelate_n_device(nint, x, old)
                torch.cuda.synchronize()
            finally:
                os.environ.pop("MI355_XE_NO_LINES", None)
            assert torch.equal(out, old), (nint, rep)


# ---- two polarisations (k_xe_i8_lines<false, 2>): 64 stations x {X, Y} = 128 rows, the reference CLI's default geometry (lib/test-clxengine.cc:66)
def _xe2(gpu, F, T):
    return gpu.clXEngine(*GPU_ARGS, False, gpu.DTYPE_BYTE, 2, 64, gpu.CLXCORR_TRIANGULAR_ORDER, 0, F, T, [])


@pytest.mark.parametrize("F,T", [(32, 32), (32, 96), (64, 64), (96, 160), (256, 32), (512, 64)])
def test_two_polarisations_bit_exact(gpu, oracle, small_units, F, T):
    """Every group type (the two diagonal groups, the six station-tile pairs), one and several K blocks, one line and many: bit exact against the oracle
    ([chan][baseline][XX, XY, YX, YY], lib/clXEngine_impl.cc:786-808) and identical to the corner-turn + correlator path."""
    import torch
    N = 64
    rng = np.random.default_rng(F + 11 * T)
    w = rng.integers(-128, 128, size=(T, N, F, 2, 2), dtype=np.int64).astype(np.int8)
    blk = _xe2(gpu, F, T)
    ref = oracle.xengine_ichar(N, F, 2, T, w.reshape(-1), exact=True)
    x = torch.from_numpy(w).cuda()
    for rep in range(2):
        out = torch.zeros(blk.get_output_buffer_size(), 2, device="cuda")
        blk.xcorrelate_device(x, out)
        torch.cuda.synchronize()
        r = blk.last_route()
        assert r["kernel"] == "k_xe_i8_lines<2 pol>", r
        assert np.array_equal(out.cpu().numpy().view(np.complex64).reshape(-1), ref), rep
    os.environ["MI355_XE_NO_LINES2"] = "1"
    try:
        old = torch.zeros_like(out)
        blk.xcorrelate_device(x, old)
        torch.cuda.synchronize()
        assert blk.last_route()["kernel"] != "k_xe_i8_lines<2 pol>"
    finally:
        os.environ.pop("MI355_XE_NO_LINES2", None)
    assert torch.equal(out, old)


def test_two_polarisations_extremes_and_host_call(gpu, oracle, small_units):
    """All -128 over 4096 frames (the combined accumulators of the diagonal pairs), alternating extremes; the host-pointer call takes the same route."""
    import torch
    N, F, T = 64, 32, 4096
    blk = _xe2(gpu, F, T)
    for fill in ("min", "alt"):
        w = np.full((T, N, F, 2, 2), -128, np.int8)
        if fill == "alt":
            w[..., 1] = 127
            w[::2, 1::2, :, 1] = np.array([127, -128], np.int8)
        ref = oracle.xengine_ichar(N, F, 2, T, w.reshape(-1), exact=True)
        out = np.empty(blk.get_output_buffer_size(), np.complex64)
        blk.xcorrelate(w.reshape(-1), out)
        assert blk.last_route()["kernel"] == "k_xe_i8_lines<2 pol>"
        assert np.array_equal(out, ref), fill


def test_two_polarisations_reference_cli_geometry(gpu):
    """64 antennas x 2 polarisations x 1024 channels x 1024 frames, the default of the reference's timing tool (lib/test-clxengine.cc:66): 256
    (line, pair group) units = one per compute unit, no corner-turn kernel; identical to the two-kernel path (which test_xengine_gpu.py pins on the oracle)."""
    import torch
    N, F, T = 64, 1024, 1024
    g = torch.Generator(device="cuda").manual_seed(8)
    x = torch.randint(-128, 128, (T, N, F, 2, 2), dtype=torch.int8, device="cuda", generator=g)
    blk = _xe2(gpu, F, T)
    out = torch.zeros(blk.get_output_buffer_size(), 2, device="cuda")
    for rep in range(2):
        out.zero_()
        blk.xcorrelate_device(x, out)
        torch.cuda.synchronize()
    r = blk.last_route()
    assert r["kernel"] == "k_xe_i8_lines<2 pol>" and r["workgroups"] == 256 and r["units_per_workgroup"] == 1, r
    os.environ["MI355_XE_NO_LINES2"] = "1"
    try:
        old = torch.zeros_like(out)
        blk.xcorrelate_device(x, old)
        torch.cuda.synchronize()
    finally:
        os.environ.pop("MI355_XE_NO_LINES2", None)
    assert torch.equal(out, old)


@pytest.mark.parametrize("F,T,nint,ng", [(32, 64, 3, 0), (64, 32, 8, 0), (256, 64, 4, 0), (64, 64, 2, 16)])
def test_two_polarisations_several_windows_per_launch(gpu, oracle, small_units, F, T, nint, ng):
    """nint windows of 64 stations x two polarisations in ONE launch (a workgroup runs its units back to back), reference layout and group-major:
    every window bit exact against the oracle."""
    import torch
    N = 64
    rng = np.random.default_rng(F + T + nint)
    wins = rng.integers(-128, 128, size=(nint, T, N, F, 2, 2), dtype=np.int64).astype(np.int8)
    blk = _xe2(gpu, F, T)
    per = blk.get_output_buffer_size()
    ref = np.concatenate([oracle.xengine_ichar(N, F, 2, T, wins[i].reshape(-1), exact=True) for i in range(nint)])
    if ng:
        W = N // ng
        x = torch.from_numpy(np.ascontiguousarray(wins.reshape(nint, T, W, ng, F, 2, 2).transpose(2, 0, 1, 3, 4, 5, 6))).cuda()
    else:
        x = torch.from_numpy(wins).cuda()
    out = torch.zeros(nint * per, 2, device="cuda")
    _run(gpu, blk, nint, x, out, ng=ng)
    r = blk.last_route()
    one = nint >= 4 or ng  # (fewer windows in the reference layout: one launch per window is faster)
    assert r["kernel"] == "k_xe_i8_lines<2 pol>" and r["launches"] == (1 if one else nint) and r["windows"] == (nint if one else 1), r
    assert np.array_equal(out.cpu().numpy().view(np.complex64).reshape(-1), ref)
