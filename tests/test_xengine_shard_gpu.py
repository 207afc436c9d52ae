"""clXEngine over several ranks of ONE process (mi355_xengine_shard_*): the corner turn between the ranks' buffers and the in-place
group-major read, bit-exact against the oracle.  One GPU is visible here, so every rank sits on device 0 and a peer copy is a device copy; the
pipeline (packing, per-destination copies, event ordering between exchange and compute streams, two slots, slab placement) is the same code
an 8-device run executes.  Unmeasured on several devices."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("W,N,F,T,npol,windows", [(1, 16, 64, 64, 1, 1), (2, 16, 64, 64, 1, 1), (4, 64, 128, 128, 1, 2), (8, 64, 256, 64, 1, 3),
                                                 (2, 16, 64, 96, 2, 2), (8, 64, 1024, 32, 1, 1)])
def test_sharded_host_call_matches_oracle(gpu, oracle, W, N, F, T, npol, windows):
    """mi355_xengine_shard_xcorrelate: `windows` windows in the reference's layout in, the reference's matrices out, every rank's slab in its
    place; then accumulate; then a second call (the other slot)."""
    rng = np.random.default_rng(W * 100 + N + windows)
    sh = gpu.clXEngineSharded([0] * W, npol, N, F, T, windows)
    per = sh.get_output_buffer_size()
    assert per == F * (N * (N + 1) // 2) * npol * npol
    for rep in range(3):
        x = rng.integers(-128, 128, size=(windows, T, N, F, npol, 2), dtype=np.int64).astype(np.int8)
        ref = np.concatenate([oracle.xengine_ichar(N, F, npol, T, x[w].reshape(-1), exact=True) for w in range(windows)])
        out = np.full(windows * per, 7 + 7j, np.complex64)
        sh.xcorrelate(x, out)
        assert np.array_equal(out, ref), rep
    sh.xcorrelate(x, out, True)
    ref2 = np.concatenate([oracle.xengine_ichar(N, F, npol, T, x[w].reshape(-1), exact=True, acc=ref[w * per:(w + 1) * per].copy()) for w in range(windows)])
    assert np.array_equal(out, ref2)
    sh.close()


def test_sharded_device_pipeline_overlaps_and_stays_exact(gpu, oracle):
    """submit_dev back to back without synchronising in between: exchange k+1 is enqueued while correlation k may still run (two slots), every
    result bit-exact; the frames of a call are produced on torch's stream right before it (mi355_xengine_shard_wait_stream orders them)."""
    import torch
    W, N, F, T, windows = 4, 64, 256, 128, 2
    rng = np.random.default_rng(5)
    sh = gpu.clXEngineSharded([0] * W, 1, N, F, T, windows)
    Ng, Fw, slab = N // W, F // W, sh.slab_items()
    rounds = 6
    xs = [rng.integers(-128, 128, size=(windows, T, N, F, 1, 2), dtype=np.int64).astype(np.int8) for _ in range(2)]
    refs = [[oracle.xengine_ichar(N, F, 1, T, x[w].reshape(-1), exact=True).reshape(F, -1) for w in range(windows)] for x in xs]
    # rank r's frames: stations [r Ng, (r+1) Ng) of every time step
    host = [[torch.from_numpy(np.ascontiguousarray(x[:, :, r * Ng:(r + 1) * Ng])) for r in range(W)] for x in xs]
    frames = [[torch.empty_like(host[0][r], device="cuda") for r in range(W)] for _ in range(rounds)]
    outs = [[torch.zeros(windows * slab, 2, device="cuda") for r in range(W)] for _ in range(rounds)]
    pinned = [[h.pin_memory() for h in hs] for hs in host]
    for k in range(rounds):
        for r in range(W):
            frames[k][r].copy_(pinned[k & 1][r], non_blocking=True)  # the producer of the frames, on torch's stream ...
            sh.wait_current_stream(r)                                # ... which the rank's compute stream waits for
        sh.submit_device(frames[k], outs[k])
    sh.synchronize()
    for k in range(rounds):
        for r in range(W):
            got = outs[k][r].cpu().numpy().view(np.complex64).reshape(windows, Fw, -1)
            for w in range(windows):
                assert np.array_equal(got[w], refs[k & 1][w][r * Fw:(r + 1) * Fw]), (k, r, w)
    sh.close()


def test_sharded_constructor_errors(gpu):
    with pytest.raises(gpu.Mi355Error):
        gpu.clXEngineSharded([0, 0, 0], 1, 64, 128, 32)      # 3 ranks do not divide 64 inputs
    with pytest.raises(gpu.Mi355Error):
        gpu.clXEngineSharded([0, 0], 1, 128, 128, 32)        # more than 64 rows: no in-place read
    with pytest.raises(gpu.Mi355Error):
        gpu.clXEngineSharded([0, 0], 1, 16, 12, 32)          # 6 channels per rank: not whole 16-byte pieces
    with pytest.raises(gpu.Mi355Error):
        gpu.clXEngineSharded([0, 99], 1, 16, 64, 32)         # no such device


@pytest.mark.parametrize("W,windows", [(1, 1), (2, 2), (4, 4), (8, 1)])
def test_sharded_streaming_host_path(gpu, oracle, W, windows):
    """acquire / submit_acquired / wait: the pinned frame slots of the handle, two exchanges in flight, results in submission order, bit exact; a
    third submit without a wait is refused (MI355_ERR_STATE), as is a wait with nothing submitted."""
    N, F, T = 64, 128, 64
    rng = np.random.default_rng(W * 10 + windows)
    sh = gpu.clXEngineSharded([0] * W, 1, N, F, T, windows)
    per = sh.get_output_buffer_size()
    assert sh.input_bytes() == windows * T * N * F * 2 and sh.pending() == 0
    with pytest.raises(gpu.Mi355Error):
        sh.wait(np.empty(windows * per, np.complex64))
    rounds = 5
    xs = [rng.integers(-128, 128, size=(windows, T, N, F, 1, 2), dtype=np.int64).astype(np.int8) for _ in range(rounds)]
    refs = [np.concatenate([oracle.xengine_ichar(N, F, 1, T, x[w].reshape(-1), exact=True) for w in range(windows)]) for x in xs]
    got = []
    out = np.empty(windows * per, np.complex64)
    for k in range(rounds):
        if sh.pending() == 2:
            sh.wait(out)
            got.append(out.copy())
        buf = sh.acquire()
        buf[:] = xs[k].reshape(-1)
        sh.submit_acquired()
    assert sh.pending() == 2
    with pytest.raises(gpu.Mi355Error):
        sh.acquire()
    while sh.pending():
        sh.wait(out)
        got.append(out.copy())
    assert len(got) == rounds
    for k in range(rounds):
        assert np.array_equal(got[k], refs[k]), k
    # the synchronous host call still works on the same handle afterwards
    sh.xcorrelate(xs[0], out)
    assert np.array_equal(out, refs[0])
    sh.close()


def test_sharded_two_polarisations(gpu, oracle, monkeypatch):
    """64 stations x two polarisations (128 rows) over two and four ranks: the receive buffer of the corner turn is read in place by the whole-line kernel
    (k_xe_i8_lines<false, 2> with stations_per_group); MI355_XE_LINES_MIN_UNITS lets the small slabs of this test take it."""
    monkeypatch.setenv("MI355_XE_LINES_MIN_UNITS", "8")
    N, F, T, windows = 64, 128, 64, 2
    rng = np.random.default_rng(21)
    for W in (2, 4):
        sh = gpu.clXEngineSharded([0] * W, 2, N, F, T, windows)
        per = sh.get_output_buffer_size()
        assert per == F * (N * (N + 1) // 2) * 4
        x = rng.integers(-128, 128, size=(windows, T, N, F, 2, 2), dtype=np.int64).astype(np.int8)
        ref = np.concatenate([oracle.xengine_ichar(N, F, 2, T, x[w].reshape(-1), exact=True) for w in range(windows)])
        out = np.empty(windows * per, np.complex64)
        sh.xcorrelate(x, out)
        assert np.array_equal(out, ref), W
        sh.acquire()[:] = x.reshape(-1)
        sh.submit_acquired()
        out2 = np.empty_like(out)
        sh.wait(out2)
        assert np.array_equal(out2, ref), W
        sh.close()


def test_sharded_two_polarisations_needs_enough_windows(gpu):
    """Without the test switch: 64 inputs x 2 polarisations over 8 ranks of 1024 channels need 8 windows per exchange to fill a device; fewer are refused at
    create with a message that says so (not by the first correlation)."""
    with pytest.raises(gpu.Mi355Error, match="at least 8 windows"):
        gpu.clXEngineSharded([0] * 8, 2, 64, 1024, 1024, 4)
    sh = gpu.clXEngineSharded([0] * 8, 2, 64, 1024, 32, 8)
    sh.close()
