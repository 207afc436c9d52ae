#!/usr/bin/env python3
"""bench.py -- throughput of the gr-clenabled hot path on MI355X.

Contract (driver):  python bench.py --gpus N --steps K --warmup W
  N>1 is launched by the driver as one rank per GPU through torch.distributed.run.
  A "step" is one pass of the headline block over one batch of synthetic input that
  is already resident in HBM: BASELINE.json configs[1], forward clFFT of 4096-point
  complex frames with a Blackman window and fftshift, 16384 frames (in+out = 1 GiB,
  beyond the 256 MiB Infinity Cache) per GPU.  Independent block instances shard one
  per GPU with no data-path collective -> "scaling": "weak".
Rank 0 prints ONE JSON line.  `value` = samples all ranks processed / max-over-ranks
time over EXACTLY K steps.  `sustained` repeats the same launch back to back for >= 2 s
(median / min of the per-launch time over batches: clocks under a long load, not a 4 ms
burst).  `roofline` prices the FFT kernel against HBM (16 algorithmic bytes per complex
sample); `cpu_baseline` times the oracle's restatement of the reference's
CPU path (clFFT_impl::testCPU) on ONE host core and on ALL of them (threads over frames)
over a bounded sample, with the host's CPU model and core counts.  `blocks` carries the
other hot-path blocks (device resident), the host-pointer work() calls (`*_hostpath`,
PCIe inclusive, never `value`), BASELINE config 1 as SURVEY 8d states it, and the
sharded X-engine (all-to-all corner turn overlapped with the correlation).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling
FFT_N = 4096
FRAMES_PER_STEP = 16384
BYTES_PER_SAMPLE = 16  # 8 B read + 8 B written per complex sample; window/twiddles are register/L2 resident


def dist_setup(ngpus):
    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != ngpus:
        raise SystemExit("bench.py --gpus %d but WORLD_SIZE=%d: launch N>1 through torch.distributed.run" % (ngpus, world))
    # MI355_BENCH_BACKEND=gloo MI355_BENCH_ONE_DEVICE=1: a dry run of the N>1 control flow with every rank on cuda:0 (a one-GPU box has no
    # second device for RCCL); the numbers of such a run mean nothing, it only shows that every rank reaches every collective
    backend = os.environ.get("MI355_BENCH_BACKEND", "nccl")
    if os.environ.get("MI355_BENCH_ONE_DEVICE"):
        local = 0
    torch.cuda.set_device(local)
    if "RANK" in os.environ:  # launched by torch.distributed.run (also with one rank, so that path is exercised)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        import datetime
        # a short collective timeout: a rank that dies in a secondary line must not hang the others for the 10-minute default
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local),
                                    timeout=datetime.timedelta(seconds=180))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world, timeout=datetime.timedelta(seconds=180))
        # The first collective builds the RCCL communicator (seconds).  Done here, the barrier that opens the timed region is a
        # plain barrier; left to that barrier, the device sits idle behind it and the K timed steps that follow run at the clocks
        # of a device that has just been idle (measured with one rank: 192 us per launch instead of 178).
        dist.barrier()
        torch.cuda.synchronize()
    return rank, world, local


def barrier(world):
    import torch
    import torch.distributed as dist
    torch.cuda.synchronize()
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
    torch.cuda.synchronize()


def time_steps(fn, steps, warmup, world):
    """W untimed steps, then exactly K steps bracketed by barrier+synchronize; returns
    (wall seconds for K steps on this rank, HIP-event seconds over the same K launches).
    ONE pair of events on the launch stream brackets the K back-to-back launches: the event figure / K is the average
    launch duration (what rocprofv3 --kernel-trace reports, plus the ~1 us hand-over between consecutive kernels).
    Events between the launches were measured to cost 7 % of the rate (each is a barrier packet the next kernel waits on)."""
    import torch
    for _ in range(warmup):
        fn()
    barrier(world)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    a.record()
    for _ in range(steps):
        fn()
    b.record()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    barrier(world)
    return wall, a.elapsed_time(b) * 1e-3


def max_over_ranks(x, world):
    return entry.load_package().shard.max_over_ranks(x)


def lowpass_taps(gain, fs, cutoff, tw, atten=53.0):
    """Hamming windowed-sinc low-pass (the definition of firdes::low_pass, lib/firdes.cc:92-137)."""
    nt = int(atten * fs / (22.0 * tw))
    nt += (nt & 1) == 0
    m = (nt - 1) // 2
    k = np.arange(-m, m + 1)
    w = 0.54 - 0.46 * np.cos(2 * np.pi * np.arange(nt) / (nt - 1))
    w0 = 2 * np.pi * cutoff / fs
    with np.errstate(invalid="ignore", divide="ignore"):
        t = np.where(k == 0, w0 / np.pi, np.sin(k * w0) / (k * np.pi)) * w
    return (t * (gain / t.sum())).astype(np.float32)


def host_cpu():
    """CPU model string, logical and physical core counts of the box the baseline runs on."""
    model, phys = "unknown", set()
    try:
        pid = cid = None
        with open("/proc/cpuinfo") as fh:
            for ln in fh:
                k, _, v = ln.partition(":")
                k, v = k.strip(), v.strip()
                if k == "model name" and model == "unknown":
                    model = v
                elif k == "physical id":
                    pid = v
                elif k == "core id":
                    cid = v
                elif not ln.strip():
                    if cid is not None:
                        phys.add((pid, cid))
                    pid = cid = None
    except OSError:
        pass
    usable = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None  # a container may hold fewer CPUs than it can see
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            q, per = fh.read().split()[:2]
            if q != "max":
                quota = float(q) / float(per)
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as fq, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fp_:
                q, per = float(fq.read()), float(fp_.read())
                if q > 0:
                    quota = q / per
        except (OSError, ValueError):
            pass
    if quota:
        usable = max(1, min(usable, int(quota + 0.5)))
    return {"cpu_model": model, "cores_total": os.cpu_count() or 1, "cores_physical": len(phys) or None, "cores_usable": usable,
            "cgroup_cpu_quota": quota}


def cpu_baseline_fft(o, window, budget_s=8.0, budget_all_s=4.0):
    """Oracle restatement of clFFT_impl::testCPU (window, float FFT, shift): ONE core (the reference's default, nthreads = 1,
    include/clenabled/clFilter.h:53), then all usable cores with the frames split over threads (ctypes releases the GIL)."""
    from concurrent.futures import ThreadPoolExecutor
    rng = np.random.default_rng(1234)
    probe = 64
    x = (rng.standard_normal(probe * FFT_N) + 1j * rng.standard_normal(probe * FFT_N)).astype(np.complex64)
    L = o.lib()

    def run(xin, yout):  # the C entry itself with caller-owned buffers: no allocation (and no page faults) inside the timed loops
        rc = L.oracle_fft_block(FFT_N, 1, window.ctypes.data, 1, o.DTYPE_COMPLEX, probe, xin.ctypes.data, yout.ctypes.data, 0)
        assert rc == 0

    y = np.empty_like(x)
    run(x, y)
    assert np.array_equal(y, o.fft_block(FFT_N, True, window, True, o.DTYPE_COMPLEX, x))
    t0 = time.perf_counter()
    run(x, y)
    per_call = time.perf_counter() - t0
    reps = max(1, int(budget_s / per_call))
    t0 = time.perf_counter()
    for _ in range(reps):
        run(x, y)
    dt = time.perf_counter() - t0
    host = host_cpu()
    nthr = max(1, host["cores_usable"])
    xs = [x.copy() for _ in range(nthr)]
    ys = [np.empty_like(x) for _ in range(nthr)]
    for k in range(nthr):
        ys[k][:] = 0  # touch the pages
    done = [0] * nthr

    budget = [0.5]

    def worker(k):  # time-bounded: every thread transforms its own 64-frame buffer until the deadline
        stop = time.perf_counter() + budget[0]
        while time.perf_counter() < stop:
            run(xs[k], ys[k])
            done[k] += probe

    with ThreadPoolExecutor(nthr) as ex:
        list(ex.map(worker, range(nthr)))  # warm-up round: threads started, clocks up
        done[:] = [0] * nthr
        budget[0] = budget_all_s
        t1 = time.perf_counter()
        list(ex.map(worker, range(nthr)))
        dta = time.perf_counter() - t1
    frames_all = sum(done)
    d = {"value": round(reps * probe * FFT_N / dt / 1e6, 2), "unit": "MSamples/s", "cores": 1, "kind": "port",
         "sample": "%d frames of %d-pt complex FFT (window+shift), oracle fft_block f32, %.1f s" % (reps * probe, FFT_N, dt),
         "all_cores": {"value": round(frames_all * FFT_N / dta / 1e6, 2), "unit": "MSamples/s", "cores": nthr,
                       "sample": "%d threads, %d frames in total, %.1f s (frames split over threads)" % (nthr, frames_all, dta)}}
    # third column ("if an FFT library is discovered ... an optional third column"): FFTW / VOLK are not in this image;
    # numpy's pocketfft is -- the same block (window multiply, single-precision transform, half swap) on one thread
    try:
        xl = x.reshape(probe, FFT_N)
        w32 = window.astype(np.float32)

        def lib_run():
            return np.fft.fftshift(np.fft.fft(xl * w32, axis=1), axes=1)

        yl = lib_run()
        if yl.dtype == np.complex64 and np.abs(yl.reshape(-1) - y).max() <= 1e-4 * np.abs(y).max():
            t2 = time.perf_counter()
            nl = 0
            while time.perf_counter() - t2 < 2.0:
                lib_run()
                nl += 1
            dl = time.perf_counter() - t2
            d["library"] = {"value": round(nl * probe * FFT_N / dl / 1e6, 2), "unit": "MSamples/s", "cores": 1, "kind": "library",
                            "what": "numpy %s pocketfft, complex64: window multiply + fft + fftshift (not the reference's FFTW; checked against the port)" % np.__version__,
                            "sample": "%d frames, %.1f s" % (nl * probe, dl)}
        else:
            d["library"] = {"skipped": "numpy.fft did not keep complex64 (%s)" % yl.dtype}
    except Exception as exc:  # noqa: BLE001
        d["library"] = {"error": "%s: %s" % (type(exc).__name__, exc)}
    d.update(host)
    return d


def config1_testcpu(o):
    """BASELINE configs[0] exactly as SURVEY 8d states it: clMathOp complex multiply, 8192-sample buffers of (1.0, 0.5),
    the reference CLI's loop -- 1 warm-up + 200 timed testCPU iterations (lib/test_clenabled.cc:1596-1650), one core."""
    n = 8192
    a = np.full(n, 1.0 + 0.5j, np.complex64)
    b = a.copy()
    c = o.mathop(o.DTYPE_COMPLEX, o.OP_MULTIPLY, a, b)
    ok = bool(np.all(c == np.complex64(0.75 + 1.0j)))
    t0 = time.perf_counter()
    for _ in range(200):
        o.mathop(o.DTYPE_COMPLEX, o.OP_MULTIPLY, a, b)
    dt = (time.perf_counter() - t0) / 200
    return {"us_per_call": round(dt * 1e6, 2), "MSamples_per_s": round(n / dt / 1e6, 1), "iterations": 200, "items": n,
            "result_is_0.75+1.0j": ok, "kind": "port (oracle mathop through ctypes; the call overhead is included)"}


def cpu_harness(seconds=1.0):
    """The oracle's CPU legs timed from C (oracle/cpu_bench: one warm-up call, then calls for `seconds` around a steady clock, one core --
    the loop of lib/test_clenabled.cc:1562-1691 without an interpreter in it).  Measurement infrastructure like the oracle itself; None
    when the binary is missing (build() makes it)."""
    import subprocess
    exe = os.path.join(ROOT, "oracle", "cpu_bench")
    if not os.path.exists(exe):
        return None
    try:
        r = subprocess.run([exe, "%.2f" % seconds], capture_output=True, text=True, timeout=120)
        return json.loads(r.stdout) if r.returncode == 0 else None
    except Exception:  # noqa: BLE001
        return None


def cpu_extras(o, o_taps):
    """Oracle (port of the reference's CPU paths) on ONE host core, ~1-2 s each, for the secondary blocks."""
    rng = np.random.default_rng(7)
    out = {}

    def timed(fn, nsamples, min_s=1.0):
        fn()
        reps, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < min_s:
            fn()
            reps += 1
        return round(reps * nsamples / (time.perf_counter() - t0) / 1e6, 2)

    n = 1 << 20
    a = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
    out["clMathOp_multiply_complex"] = timed(lambda: o.mathop(o.DTYPE_COMPLEX, o.OP_MULTIPLY, a, a), n)   # clMathOp_impl::testCPU
    out["clMathConst_multiply_complex"] = timed(lambda: o.mathconst(o.DTYPE_COMPLEX, o.OP_MULTIPLY, 2.0, a), n)
    taps65, taps2048 = o_taps
    m = 192 * 1024
    f = o.FFTFilter(1, taps65)
    out["clFilter_fft_65taps"] = timed(lambda: f.filter(m, a[:m]), m)                                       # fft_filter_ccf::filter
    out["clFilter_fir_65taps"] = timed(lambda: o.fir_ccf(taps65, a[:m + 64], m), m)                         # fir_filter_ccf::filterN
    ct = (taps65 * np.exp(1j * np.pi * np.arange(65) / 8)).astype(np.complex64)
    out["clComplexFilter_fft_65ctaps"] = timed(lambda: o.fir_ccc(ct, a[:m + 64], m), m)                     # fir_filter_ccc (the reference has no FFT mode for complex taps)
    buf = 65536
    out["clPolyphaseChannelizer_64x32_stream"] = timed(lambda: o.pfb(taps2048, buf, 64, 64, list(range(64)), a[:buf + 2048 - 64]), buf)
    N, F, T = 64, 8, 1024  # 8 of the 1024 channels (channels are independent: the rate per sample is the same); the key says so
    x8 = rng.integers(-127, 128, size=T * N * F * 2, dtype=np.int64).astype(np.int8)
    out["clXEngine_64ant_1024ch_1024t_ichar"] = {"MSamples_per_s": timed(lambda: o.xengine_ichar(N, F, 1, T, x8, exact=False), N * F * T),  # kernel text restated
                                                 "sample": "64 antennas x 8 channels x 1024 frames per call (8 of the 1024 channels)"}
    return out


def sustained(fn, samples_per_launch, min_s=2.0, batch=64):
    """>= min_s of back-to-back launches in batches of `batch`, one event pair per batch (events between single launches cost
    7 % of the rate): whole-leg rate, median and min of the per-launch time over the batches."""
    import torch
    torch.cuda.synchronize()
    evs = []
    t0 = time.perf_counter()
    launches = 0
    while True:
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(batch):
            fn()
        b.record()
        evs.append((a, b))
        launches += batch
        if len(evs) % 8 == 0:           # bound the queue depth: wait for the batch issued eight batches ago
            evs[-8][1].synchronize()
            if time.perf_counter() - t0 >= min_s:
                break
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0

    def finish():
        # (read out afterwards: ~200 event queries are milliseconds of host time during which the device would sit idle
        # right before the K timed steps)
        per = sorted(a.elapsed_time(b) * 1e3 / batch for a, b in evs)  # us per launch
        return {"seconds": round(wall, 2), "launches": launches, "MSamples_per_s": round(launches * samples_per_launch / wall / 1e6, 1),
                "us_per_launch_median": round(per[len(per) // 2], 2), "us_per_launch_min": round(per[0], 2),
                "us_per_launch_max": round(per[-1], 2),
                "hbm_frac_median": round(samples_per_launch * BYTES_PER_SAMPLE / (per[len(per) // 2] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)}
    return finish


def hostpath_blocks(pkg, o_taps, dev):
    """Host-pointer work() calls at the reference's call sizes (what the GNU Radio scheduler hands a block): pageable numpy
    buffers in, pageable out, PCIe both ways, blocking like the reference's enqueueReadBuffer.  Secondary, never `value`."""
    rng = np.random.default_rng(11)
    args = (1, 2, 0, dev)
    out = {}

    def lat(fn, iters=200, warm=20):
        for _ in range(warm):
            fn()
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            for _ in range(iters // 5):
                fn()
            ts.append((time.perf_counter() - t0) / (iters // 5))
        ts.sort()
        return ts[len(ts) // 2], ts[0]

    def entry_(n, med, mn):
        return {"us_per_call_median": round(med * 1e6, 1), "us_per_call_min": round(mn * 1e6, 1), "items_per_call": n,
                "MSamples_per_s": round(n / med / 1e6, 1)}

    def crandn(n):
        return (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)

    n = 8192
    a, b = crandn(n), crandn(n)
    c = np.empty_like(a)
    mul = pkg.clMathOp(pkg.DTYPE_COMPLEX, *args, pkg.MATHOP_MULTIPLY)
    out["clMathOp_8192_hostpath"] = entry_(n, *lat(lambda: mul.work(n, [a, b], [c])))
    x = crandn(4096)
    y = np.empty_like(x)
    fft = pkg.clFFT(4096, pkg.CLFFT_FORWARD, np.blackman(4096).astype(np.float32), pkg.DTYPE_COMPLEX, *args, 0, 1, True)
    out["clFFT_4096_1vector_hostpath"] = entry_(4096, *lat(lambda: fft.work(1, [x], [y])))
    n = 32768
    xf = crandn(n + 64)
    yf = np.empty(n, np.complex64)
    flt = pkg.clFilter(*args, 1, o_taps[0], 1, 0, False)
    out["clFilter_fft_65taps_32768_hostpath"] = entry_(n, *lat(lambda: flt.work(n, [xf], [yf])))
    buf = 65536
    pfb = pkg.clPolyphaseChannelizer(*args, o_taps[1], buf, 64, 64, list(range(64)))
    xp = crandn(pfb.ninput())
    yp = np.empty(pfb.noutput(), np.complex64)
    out["clPolyphaseChannelizer_64x32_buf65536_hostpath"] = entry_(buf, *lat(lambda: pfb.general_work(buf, None, [xp], [yp]), iters=100))
    # large streaming call: pinned double-buffered chunks, H2D of chunk c+1 under the kernel / D2H of chunk c
    n = 1 << 24
    xl = crandn(n)
    yl = np.empty_like(xl)
    med, mn = lat(lambda: fft.work(n // 4096, [xl], [yl]), iters=5, warm=1)
    out["clFFT_4096_2p24_samples_hostpath"] = entry_(n, med, mn)
    # X-engine: whole integration windows through the double-buffered submit()/wait() pipeline (host copy into the pinned
    # slot + H2D + correlation + D2H; the frame gather of general_work() is the C++ CLI's --xengine-e2e figure)
    N, F, T = 64, 1024, 1024
    xe = pkg.clXEngine(*args, False, pkg.DTYPE_BYTE, 1, N, 1, 0, F, T, [])
    xi = rng.integers(-127, 128, size=T * N * F * 2, dtype=np.int64).astype(np.int8)
    vis = np.empty(xe.get_output_buffer_size(), np.complex64)
    xe.xcorrelate(xi, vis)
    xe.xcorrelate(xi, vis)  # (two warm-up calls: the second one brings up the second staging slot -- 128 MiB of pinned memory -- outside the timed loop)
    k = 6
    t0 = time.perf_counter()
    xe.submit(xi)
    for _ in range(k - 1):
        xe.submit(xi)
        xe.wait(vis)
    xe.wait(vis)
    dt = (time.perf_counter() - t0) / k
    out["clXEngine_64ant_1024ch_1024t_ichar_hostpath"] = {"us_per_integration": round(dt * 1e6, 1), "integrations": k,
                                                          "total_input_MSamples_per_s": round(N * F * T / dt / 1e6, 1),
                                                          "input_Gbit_per_s": round(N * F * T * 16 / dt / 1e9, 1)}
    del xe
    # the whole block path of the C++ layer: general_work()-style calls of the CLI (frame gather of every input stream into the pinned
    # window, asynchronous correlation, result delivery) -- the counterpart of what lib/test-clxengine.cc:309-332 times
    try:
        import re
        import subprocess
        cli = os.path.join(ROOT, "gr-clenabled_amd", "test-clenabled-mi355")
        r = subprocess.run([cli, "--device=%d" % dev, "--xengine-e2e=4"], capture_output=True, text=True, timeout=120)
        rows = {}
        for m in re.finditer(r"(\d+) frames? per work_test call: ([0-9.]+) ms per integration", r.stdout):
            rows["frames_per_call_%s" % m.group(1)] = {"ms_per_integration": float(m.group(2)),
                                                        "total_input_MSamples_per_s": round(N * F * T / (float(m.group(2)) * 1e-3) / 1e6, 1)}
        rows["checked"] = r.stdout.strip().endswith("ok")
        out["clXEngine_e2e_gather_hostpath"] = rows
    except Exception as exc:  # noqa: BLE001
        out["clXEngine_e2e_gather_hostpath"] = {"error": "%s: %s" % (type(exc).__name__, exc)}
    return out


def gather_floats(value, world):
    """Every rank's float, in rank order (per-GPU lines of the replica-sharded blocks)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or world == 1:
        return [float(value)]
    t = torch.tensor([float(value)], dtype=torch.float64, device="cuda")
    outs = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(outs, t)
    return [float(o.item()) for o in outs]


def extra_blocks(pkg, o_taps, dev, steps, warmup, world, rank, out=None):
    """Secondary lines: the other hot-path blocks, device resident, same timing method
    (per-GPU figures of this rank; the headline above carries the multi-GPU aggregate).  Results go into `out` as they are
    measured, so whatever exists survives a failure or the watchdog."""
    import torch
    out = {} if out is None else out
    args = (1, 2, 0, dev)

    def rate(fn, nsamples, bytes_per_sample, extra=None):
        # at least ~60 ms of back-to-back launches per row (a 5-launch burst right after an idle gap runs at lower clocks: the
        # same kernels measured 10-15 % slower that way); the launch count is fixed from one probe launch, identically on every rank
        _, probe = time_steps(fn, 1, 3, world)
        n_steps = int(min(400, max(steps, 0.06 / max(probe, 1e-6))))
        n_steps = int(max_over_ranks(float(n_steps), world))
        _, ev = time_steps(fn, n_steps, warmup, world)
        dt = ev / n_steps
        d = {"MSamples_per_s": round(nsamples / dt / 1e6, 1), "us_per_launch": round(dt * 1e6, 2),
             "GBps": round(nsamples * bytes_per_sample / dt / 1e9, 1),
             "hbm_frac": round(nsamples * bytes_per_sample / dt / 1e9 / HBM_PEAK_GBS, 4)}
        if extra:
            d.update(extra(dt))
        return d

    def rotate(make, bytes_each, call):
        # A launch whose whole input is smaller than the 256 MiB Infinity Cache would otherwise be served from it on every repeat of the same
        # buffer (config 5: 54 us on one buffer, 86 us when the input really comes from HBM -- DESIGN section 7c): the X-engine rows walk over
        # enough distinct inputs (>= 640 MB in total) that every launch reads HBM, as a stream of new integrations does.
        k = max(2, int(-(-640e6 // bytes_each)) + 1)
        bufs = [make() for _ in range(k)]
        state = [0]

        def fn():
            call(bufs[state[0] % k])
            state[0] += 1
        return fn, bufs

    n = 1 << 26  # 512 MiB per buffer: past the 256 MiB Infinity Cache
    a = torch.randn(n, 2, device="cuda")
    b = torch.randn(n, 2, device="cuda")
    c = torch.empty_like(a)
    mul = pkg.clMathOp(pkg.DTYPE_COMPLEX, *args, pkg.MATHOP_MULTIPLY)
    out["clMathOp_multiply_complex"] = rate(lambda: mul.work_device(n, [a, b], [c]), n, 24)
    mc = pkg.clMathConst(pkg.DTYPE_COMPLEX, *args, 2.0, pkg.MATHOP_MULTIPLY)
    out["clMathConst_multiply_complex"] = rate(lambda: mc.work_device(n, [a], [c]), n, 16)
    del b
    # BASELINE configs[2]: low-pass FFT filter, 65 taps, decim 1 (device-resident stream of 2^26 samples)
    taps65, taps2048 = o_taps
    nf = n - 64
    flt = pkg.clFilter(*args, 1, taps65, 1, 0, False)
    out["clFilter_fft_65taps"] = rate(lambda: flt.work_device(nf, [a], [c]), nf, 16)
    out["clFilter_fft_65taps"]["fft_size"] = flt.fftsize()
    fir = pkg.clFilter(*args, 1, taps65, 1, 0, True)
    # direct form: 4 flop per tap and sample on the fp32 matrix cores -> compute bound well before HBM (157 TFLOP/s fp32 peak, vector or matrix)
    out["clFilter_fir_65taps"] = rate(lambda: fir.work_device(nf, [a], [c]), nf, 16,
                                      lambda dt: {"TFLOPs": round(4.0 * 65 * nf / dt / 1e12, 1), "mfma_frac_f32_157TF": round(4.0 * 65 * nf / dt / 157e12, 3),
                                                  "bound": "mfma"})
    ct = (taps65 * np.exp(1j * np.pi * np.arange(65) / 8)).astype(np.complex64)
    cfl = pkg.clComplexFilter(*args, 1, ct, 1, 0, use_time=False)
    out["clComplexFilter_fft_65ctaps"] = rate(lambda: cfl.work_device(nf, [a], [c]), nf, 16)
    # a long filter (3000 taps): partitioned fast convolution, two segments summed in the frequency domain per block
    rng3 = np.random.default_rng(3000)
    taps3000 = (rng3.standard_normal(3000) / np.sqrt(3000.0)).astype(np.float32)
    nl = n - 3000
    lfl = pkg.clFilter(*args, 1, taps3000, 1, 0, False)
    out["clFilter_fft_3000taps"] = rate(lambda: lfl.work_device(nl, [a], [c]), nl, 16)
    # other transform sizes of the headline block: one smaller, the one-pass 16384 / 32768 kernels, the two-kernel and the
    # three-pass workspace schemes
    for fn_ in (1024, 16384, 32768, 65536, 131072, 1048576):
        fb = pkg.clFFT(fn_, pkg.CLFFT_FORWARD, np.blackman(fn_).astype(np.float32), pkg.DTYPE_COMPLEX, *args, 0, 1, True)
        nv = n // fn_
        out["clFFT_%d" % fn_] = rate(lambda: fb.work_device(nv, [a], [c]), nv * fn_, 16)
    # lengths that are not a power of two: 2^a 3^b 5^c 7^d through the mixed-radix kernel (one pass over HBM), a prime through chirp-z
    for fn_ in (1000, 3000, 12000, 4099):
        fb = pkg.clFFT(fn_, pkg.CLFFT_FORWARD, np.blackman(fn_).astype(np.float32), pkg.DTYPE_COMPLEX, *args, 0, 1, True)
        nv = (n // 2) // fn_
        path = "mixed radix" if fn_ != 4099 else "chirp-z (prime length)"
        out["clFFT_%d" % fn_] = rate(lambda: fb.work_device(nv, [a], [c]), nv * fn_, 16, lambda dt, path=path: {"path": path})
    # BASELINE configs[3]: polyphase channelizer 64 ch x 32 taps/arm; streaming buffer and the 65536-item call
    for buf, key in (((1 << 26) - (1 << 16), "clPolyphaseChannelizer_64x32_stream"), (65536, "clPolyphaseChannelizer_64x32_buf65536")):
        pfb = pkg.clPolyphaseChannelizer(*args, taps2048, buf, 64, 64, list(range(64)))
        assert pfb.ninput() <= a.shape[0] and pfb.noutput() <= c.shape[0]
        xi = a[:pfb.ninput()]
        yo = c[:pfb.noutput()]
        out[key] = rate(lambda: pfb.work_device([xi], [yo]), buf, 16)
        if buf == 65536:
            # The eager figure above is the HOST's launch rate (python + ctypes + hipLaunchKernel per 65536-item call), not the
            # device's.  Device side: the same launch replayed 256 x from a HIP graph; and what general_work() does when the
            # scheduler offers several output multiples -- nbuf buffers of the stream in ONE launch (work_device(nbuf=8)).
            try:
                gr_ = torch.cuda.CUDAGraph()
                side = torch.cuda.Stream()
                with torch.cuda.stream(side):
                    with torch.cuda.graph(gr_, stream=side):
                        for _ in range(256):
                            pfb.work_device([xi], [yo])
                torch.cuda.synchronize()
                _, evg = time_steps(gr_.replay, max(5, steps), 2, world)
                out[key]["us_per_launch_graph_replay"] = round(evg / max(5, steps) / 256 * 1e6, 2)
            except Exception as exc:  # noqa: BLE001
                out[key]["graph_replay_error"] = "%s: %s" % (type(exc).__name__, exc)
            nb = 8
            xb = a[:nb * buf + 2048 - 64]
            yb = c[:nb * buf]
            _, evb = time_steps(lambda: pfb.work_device([xb], [yb], nbuf=nb), steps, warmup, world)
            out[key]["batched_x8_us_per_buffer"] = round(evb / steps / nb * 1e6, 2)
            out[key]["batched_x8_MSamples_per_s"] = round(nb * buf * steps / evb / 1e6, 1)
        # BASELINE configs[3] is "8 independent instances, 1 per GPU": every rank's own rate, no collective in the data path
        out[key]["per_gpu_MSamples_per_s"] = [round(v, 1) for v in gather_floats(out[key]["MSamples_per_s"], world)]
    # frequency-domain cross-correlator, 4 time-series inputs of 1024-point vectors (reference + 3):
    # algorithmic bytes per input sample = 8 read + 4 written per non-reference input
    xn, xin = 1024, 4
    xfr = (n // xin) // xn
    xc = pkg.clxcorrelate_fft_vcf(xn, xin, *args, 2)
    xi = [a[i * xfr * xn:(i + 1) * xfr * xn] for i in range(xin)]
    xo = [c.view(-1)[i * xfr * xn:(i + 1) * xfr * xn] for i in range(xin - 1)]
    out["clxcorrelate_fft_vcf_1024_4in_timeseries"] = rate(lambda: xc.work_device(xfr, xi, xo), xin * xfr * xn, (8 * xin + 4 * (xin - 1)) / xin)
    # X-engine on complex-float input (fp32 matrix cores), same shape as configs[4]
    Nc, Fc, Tc = 64, 1024, 1024
    if a.shape[0] * 2 >= Tc * Nc * Fc * 2:
        xf = a.view(-1)[:Tc * Nc * Fc * 2]
        xec = pkg.clXEngine(*args, False, pkg.DTYPE_COMPLEX, 1, Nc, 1, 0, Fc, Tc, [])
        visc = torch.zeros(xec.get_output_buffer_size(), 2, device="cuda")
        flopc = 8.0 * Fc * (Nc * (Nc + 1) // 2) * Tc
        rc = rate(lambda: xec.xcorrelate_device(xf, visc), Nc * Fc * Tc, 8,
                  lambda dt: {"TFLOPs": round(flopc / dt / 1e12, 1), "mfma_frac_f32_157TF": round(flopc / dt / 157.3e12, 4)})
        out["clXEngine_64ant_1024ch_1024t_cf32"] = rc
        del xec, visc
        # off the tuned geometries (found by sweeping the parameters, DESIGN_EXPERIMENTS A.3b / A.5b / A.6b): a channel count whose rows are not
        # whole 128-byte lines, a channelizer with 100 channels, a time-domain filter decimating by 16
        Fr = 1000
        xf = a.view(-1)[:Tc * Nc * Fr * 2]
        xec = pkg.clXEngine(*args, False, pkg.DTYPE_COMPLEX, 1, Nc, 1, 0, Fr, Tc, [])
        visc = torch.zeros(xec.get_output_buffer_size(), 2, device="cuda")
        out["clXEngine_64ant_1000ch_1024t_cf32"] = rate(lambda: xec.xcorrelate_device(xf, visc), Nc * Fr * Tc, 8)
        del xec, visc
    Mc = 100
    tp100 = np.resize(taps2048, Mc * 32).astype(np.float32)
    bufc = ((n // 2) // Mc) * Mc
    pfbc = pkg.clPolyphaseChannelizer(*args, tp100, bufc, Mc, Mc, list(range(Mc)))
    xi = a[:pfbc.ninput()]
    yo = c[:pfbc.noutput()]
    out["clPolyphaseChannelizer_100x32_stream"] = rate(lambda: pfbc.work_device([xi], [yo]), bufc, 16)
    del pfbc
    # 512 channels x 32 taps per arm: the ring kernel with one 512-thread workgroup per CU (round 6; two kernels before)
    tp512 = np.resize(taps2048, 512 * 32).astype(np.float32)
    buf5 = ((n // 2) // 512) * 512
    pfb5 = pkg.clPolyphaseChannelizer(*args, tp512, buf5, 512, 512, list(range(512)))
    xi = a[:pfb5.ninput()]
    yo = c[:pfb5.noutput()]
    out["clPolyphaseChannelizer_512x32_stream"] = rate(lambda: pfb5.work_device([xi], [yo]), buf5, 16)
    del pfb5
    fd = pkg.clFilter(*args, 16, taps65, 1, 0, True)
    nd = (n - 64) // 16
    out["clFilter_fir_65taps_decim16"] = rate(lambda: fd.work_device(nd, [a], [c]), nd * 16, 8 + 0.5)
    # a decimation above the filter length: the per-output kernel reads only the 65 samples an output needs (35 of every 100 are never
    # touched), so the rate is quoted in input samples consumed per second and carries no HBM fraction
    fd = pkg.clFilter(*args, 100, taps65, 1, 0, True)
    nd = (n - 64) // 100
    rr = rate(lambda: fd.work_device(nd, [a], [c]), nd * 100, 8 * 0.65 + 0.08)
    rr.pop("hbm_frac", None)
    rr["GBps"] = round(rr["GBps"], 1)
    rr["note"] = "GBps = bytes the outputs need (65 of every 100 input samples + the outputs)"
    out["clFilter_fir_65taps_decim100"] = rr
    del fd
    del a, c
    torch.cuda.empty_cache()
    # BASELINE configs[4]: X-engine 64 antennas x 1024 channels x 1024 frames, IChar.  Both fractions.
    N, F, T = 64, 1024, 1024
    Fw = F // world  # channel slab of this rank after the corner turn (gr-clenabled_amd/shard.py)
    g = torch.Generator(device="cuda").manual_seed(42 + rank)
    xe = pkg.clXEngine(*args, False, pkg.DTYPE_BYTE, 1, N, 1, 0, Fw, T, [])
    x8 = torch.randint(-127, 128, (T, N, Fw, 1, 2), dtype=torch.int8, device="cuda", generator=g)
    vis = torch.zeros(xe.get_output_buffer_size(), 2, device="cuda")
    nb = N * (N + 1) // 2
    flop = 8.0 * Fw * nb * T
    alg_bytes = x8.numel() + vis.numel() * 4

    def xe_extra(dt):
        # plus the reference tool's own figures of merit (lib/test-clxengine.cc:287-300): total input
        # samples/s, per-stream rate, input bits/s
        return {"TFLOPs": round(flop / dt / 1e12, 1), "mfma_frac_i8_5POPS": round(flop / dt / 5e15, 4),
                "hbm_frac_algorithmic": round(alg_bytes / dt / 1e9 / HBM_PEAK_GBS, 4), "channels_this_rank": Fw,
                "per_stream_MSamples_per_s": round(Fw * T / dt / 1e6, 1), "input_Gbit_per_s": round(N * Fw * T * 16 / dt / 1e9, 1)}

    mk = lambda: torch.randint(-127, 128, (T, N, Fw, 1, 2), dtype=torch.int8, device="cuda", generator=g)
    fn_rot, bufs = rotate(mk, x8.numel(), lambda x: xe.xcorrelate_device(x, vis))
    r = rate(fn_rot, N * Fw * T, 2, xe_extra)
    r.pop("hbm_frac", None)
    rt = xe.last_route()  # which kernel that was (mi355_xengine_last_route): the route depends on geometry, alignment and environment
    r["kernel"] = rt["kernel"]
    r["route"] = {k: rt[k] for k in ("workgroups", "tsplit", "in_launch_reduce", "touches")}
    r["inputs"] = "%d distinct windows in rotation (%.0f MB): every launch reads its input from HBM" % (len(bufs), len(bufs) * x8.numel() / 1e6)
    del bufs, fn_rot
    # the same call on ONE buffer over and over (what rounds 1-3 quoted): an input below 256 MiB stays in the Infinity Cache between launches
    rs_ = rate(lambda: xe.xcorrelate_device(x8, vis), N * Fw * T, 2)
    r["same_buffer_us_per_launch"] = rs_["us_per_launch"]
    r["same_buffer_note"] = "one %.0f MB input re-read every launch (Infinity-Cache resident): not the headline, kept for comparison with rounds 1-3" % (x8.numel() / 1e6)
    out["clXEngine_64ant_1024ch_1024t_ichar"] = r
    t_batched8 = None
    if world > 1:
        # SURVEY 8e (B), the comparison form of the sharding: every rank ingests ALL antennas of its F/W channels (the corner turn done by
        # the network in front of the GPUs, as packet-switched FX correlators do) -- no data-path collective at all; a rank's slab is too
        # small to fill the device one window at a time, so eight windows go into one launch (mi355_xengine_xcorrelate_n_dev)
        # -- 8 windows per launch, and as many as give every CU a whole integration of a 32-byte slice (no time ranges, no partial sums)
        units = max(1, Fw * 2 // 32)
        nfill = max(8, (-(-256 // units) + 7) // 8 * 8)
        per_nint = {}
        for nint in sorted({8, nfill}):
            vb = torch.zeros(nint * xe.get_output_buffer_size(), 2, device="cuda")
            fn_rot, bufs = rotate(lambda: torch.randint(-127, 128, (nint, T, N, Fw, 1, 2), dtype=torch.int8, device="cuda", generator=g),
                                  nint * T * N * Fw * 2, lambda x: xe.xcorrelate_n_device(nint, x, vb))
            rb = rate(fn_rot, nint * N * Fw * T, 2)
            per_nint[nint] = max_over_ranks(rb["us_per_launch"], world) / nint
            del bufs, fn_rot, vb
            torch.cuda.empty_cache()
        nint = min(per_nint, key=per_nint.get)
        tw = per_nint[nint]
        # the N = 1 numbers IN THIS LINE: the whole 64 x 1024 x 1024 integration on one device (every rank measures it on its own GPU at the
        # same time; the slowest is quoted), one window per call and eight per call.  The sharded rows carry ONE efficiency figure,
        # t(1 GPU, eight windows per launch) / (N x t(N GPUs, eight windows per launch)) -- like for like; the single call's time is quoted
        # beside it as a time, not as a ratio (a batched N-GPU run against an unbatched one-GPU call says 1.4 at N = 1)
        xf = pkg.clXEngine(*args, False, pkg.DTYPE_BYTE, 1, N, 1, 0, F, T, [])
        v1 = torch.zeros(8 * xf.get_output_buffer_size(), 2, device="cuda")
        fn_rot, bufs = rotate(lambda: torch.randint(-127, 128, (T, N, F, 1, 2), dtype=torch.int8, device="cuda", generator=g), T * N * F * 2,
                              lambda x: xf.xcorrelate_device(x, v1))
        r1 = rate(fn_rot, N * F * T, 2)
        t1 = max_over_ranks(r1["us_per_launch"], world)
        del bufs, fn_rot
        fn_rot, bufs = rotate(lambda: torch.randint(-127, 128, (8, T, N, F, 1, 2), dtype=torch.int8, device="cuda", generator=g), 8 * T * N * F * 2,
                              lambda x: xf.xcorrelate_n_device(8, x, v1))
        r1b = rate(fn_rot, 8 * N * F * T, 2)
        t1b = max_over_ranks(r1b["us_per_launch"], world) / 8
        del xf, bufs, fn_rot, v1
        torch.cuda.empty_cache()
        out["clXEngine_n1_reference"] = {"us_per_integration_one_gpu": round(t1, 2), "us_per_window_one_gpu_8_windows_per_launch": round(t1b, 2),
                                         "what": "the full 64 ant x 1024 ch x 1024 frame integration on ONE device, "
                                         "measured by every rank of this run on its own GPU (slowest rank), inputs in rotation (read from HBM)"}
        out["clXEngine_channel_sharded"] = {"us_per_window_all_ranks": round(tw, 2), "windows_per_launch": nint, "channels_per_rank": Fw,
                                            "us_per_window_by_windows_per_launch": {str(k): round(v, 2) for k, v in per_nint.items()},
                                            "total_input_MSamples_per_s": round(N * F * T / tw, 1), "n_gpus": world,
                                            "n1_us_per_integration_single_call": round(t1, 2),
                                            "scaling_efficiency_vs_n1_batched": round(t1b / (world * tw), 3), "n1_us_per_window_batched": round(t1b, 2),
                                            "collective": "none (every rank ingests all antennas of its F/W channels)"}
    if world == 1:
        # A stream of integrations handed over 4 ... 32 at a time (mi355_xengine_xcorrelate_n_dev): every unit is a whole integration of one line's
        # pair group (no time ranges, no partial sums), 256 ... 2048 units on 256 CUs.  Inputs in rotation (every launch reads HBM).
        row = {}
        def lines_kernel(units, cus=256, max_items=64):  # mi355_xe_lines_ok's rule for this geometry (csrc/xengine_lines.hip)
            if os.environ.get("MI355_XE_NO_LINES") or N != 64 or Fw % 64 or units < cus or units % 32:
                return False
            for it in range(-(-units // cus), max_items + 1):
                if units % it == 0 and (units // it) % 32 == 0:
                    return (units // it) * 8 >= cus * 7
            return False
        for nint in (4, 8, 16, 32):
            vb = torch.zeros(nint * xe.get_output_buffer_size(), 2, device="cuda")
            fn_rot, bufs = rotate(lambda: torch.randint(-127, 128, (nint, T, N, Fw, 1, 2), dtype=torch.int8, device="cuda", generator=g),
                                  nint * T * N * Fw * 2, lambda x: xe.xcorrelate_n_device(nint, x, vb))
            rb = rate(fn_rot, nint * N * Fw * T, 2)
            tw = rb["us_per_launch"] / nint
            # (which kernel: mi355_xe_lines_ok -- 64 stations, whole-line rows, enough (window, line, pair group) units to fill the device in equal
            # shares of at most 64 per workgroup: the whole-line kernel of csrc/xengine_lines.hip; otherwise the 32-byte-slice kernel)
            units = nint * (Fw // 64) * 4
            lines = xe.last_route()["kernel"] == "k_xe_i8_lines"
            assert lines == lines_kernel(units), (xe.last_route(), units)
            row["windows_per_launch_%d" % nint] = {"us_per_window": round(tw, 2), "MSamples_per_s": rb["MSamples_per_s"],
                                                  "hbm_frac_algorithmic": round(alg_bytes / (tw * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                                                  "distinct_inputs_in_rotation": len(bufs),
                                                  "kernel": "k_xe_i8_lines" if lines else "k_xe_i8_fused"}
            del bufs, fn_rot, vb
            torch.cuda.empty_cache()
        out["clXEngine_64ant_1024ch_1024t_ichar_batched"] = row
        t_batched8 = row["windows_per_launch_8"]["us_per_window"]
    del xe, x8, vis
    # The per-rank problem of the 8-GPU antenna-group sharding: after the corner turn a rank correlates 64 antennas x 128
    # channels.  One window per launch cannot fill the device; the batched entry point (mi355_xengine_xcorrelate_n_dev, what one
    # all-to-all over `windows` integrations feeds) can.  compute_ratio_vs_one_gpu_batched = t(1024 channels, 8 windows per launch) /
    # (8 x t(128 channels)): what eight ranks' CORRELATIONS alone would scale to -- the exchange is not in it; xgmi_floor_us_per_window is the
    # time one window's 2 MiB per peer needs on one 153 GB/s link, which the antenna-group form cannot beat.
    if world == 1:
        Fr = 128
        xr = pkg.clXEngine(*args, False, pkg.DTYPE_BYTE, 1, N, 1, 0, Fr, T, [])
        perw = xr.get_output_buffer_size()
        row = {"channels": Fr, "antennas": N, "frames": T, "one_gpu_us_per_window_8_windows_per_launch": t_batched8,
               "xgmi_floor_us_per_window": round(T * (N // 8) * Fr * 2 / 153e9 * 1e6, 2),
               "xgmi_floor_note": "2 MiB per peer and window over one 153 GB/s link; the exchange runs under the previous batch's correlation, so an "
                                  "8-GPU antenna-group run is bounded by max(this, us_per_window), not by us_per_window alone"}
        for nint in (1, 8, 32):
            vb = torch.zeros(nint * perw, 2, device="cuda")
            fn_rot, bufs = rotate(lambda: torch.randint(-127, 128, (nint, T, N, Fr, 1, 2), dtype=torch.int8, device="cuda", generator=g),
                                  nint * T * N * Fr * 2, lambda x: xr.xcorrelate_n_device(nint, x, vb))
            rr = rate(fn_rot, nint * N * Fr * T, 2)
            tw = rr["us_per_launch"] / nint
            row["windows_per_launch_%d" % nint] = {"us_per_launch": rr["us_per_launch"], "us_per_window": round(tw, 2),
                                                  "distinct_inputs_in_rotation": len(bufs)}
            if t_batched8:
                row["windows_per_launch_%d" % nint]["compute_ratio_vs_one_gpu_batched"] = round(t_batched8 / (8 * tw), 3)
            del bufs, fn_rot, vb
        out["clXEngine_perrank_64ant_128ch_1024t_ichar"] = row
        del xr
        # Large arrays (more than 64 rows): corner turn + the persistent one-pass correlator k_xe_corr_sb -- except 64 stations x two polarisations
        # (the reference CLI's default), which the whole-line kernel takes in one pass (k_xe_i8_lines<2 pol>, round 6).  Compute-bound by the
        # algorithm's count (2 x rows operations per input byte), so the int8 matrix-core fraction is the yardstick; the two-kernel
        # form moves input + tiles written + tiles read + output, which is what bounds it.
        for (Na, Fa, npa, key) in ((128, 1024, 1, "clXEngine_128ant_1024ch_1024t_ichar"), (64, 1024, 2, "clXEngine_64ant_dualpol_1024ch_1024t_ichar"),
                                   (256, 512, 1, "clXEngine_256ant_512ch_1024t_ichar")):
            xl = pkg.clXEngine(*args, False, pkg.DTYPE_BYTE, npa, Na, 1, 0, Fa, T, [])
            vl = torch.zeros(xl.get_output_buffer_size(), 2, device="cuda")
            in_bytes = T * Na * Fa * npa * 2
            fn_rot, bufs = rotate(lambda: torch.randint(-127, 128, (T, Na, Fa, npa, 2), dtype=torch.int8, device="cuda", generator=g), in_bytes,
                                  lambda x: xl.xcorrelate_device(x, vl))
            ops = 8.0 * Fa * (Na * (Na + 1) // 2) * T * npa * npa
            algb = in_bytes + vl.numel() * 4
            moved = 3 * in_bytes + vl.numel() * 4  # input read, tiles written, tiles read, output written
            rl = rate(fn_rot, Na * npa * Fa * T, 2,
                      lambda dt: {"TOPs": round(ops / dt / 1e12, 1), "mfma_frac_i8_5POPS": round(ops / dt / 5e15, 4),
                                  "hbm_frac_algorithmic": round(algb / dt / 1e9 / HBM_PEAK_GBS, 4),
                                  "hbm_frac_bytes_moved": round(moved / dt / 1e9 / HBM_PEAK_GBS, 4)})
            rl.pop("hbm_frac", None)
            rl.pop("GBps", None)
            rl["kernel"] = xl.last_route()["kernel"]
            if rl["kernel"].startswith("k_xe_i8_lines"):  # one pass over the input in the reference layout: no tiles written or read back
                rl.pop("hbm_frac_bytes_moved", None)
            out[key] = rl
            del xl, bufs, fn_rot, vl
            torch.cuda.empty_cache()
    return out


def sharded_xengine(pkg, dev, steps, world, rank, windows=8):
    """BASELINE configs[4] sharded the way SURVEY 8e describes: every rank ingests its antenna group, ONE all-to-all corner
    turn (RCCL over xGMI, gr-clenabled_amd/shard.py) hands every rank its channel slice of all antennas, then the local
    correlation.  An exchange carries `windows` integration windows and the rank correlates them in ONE launch
    (mi355_xengine_xcorrelate_n_dev): a rank's slice alone (128 channels at 8 GPUs) cannot fill a device.  The send blocks are
    packed by one strided device copy, the receive buffer is read in place by the fused kernel (stations_per_group), and
    exchange i+1 runs on a side stream under the correlation of i.  Timed with HIP events on the launch stream; the bare
    correlation call of the same batch is timed beside it, the difference is what the pipeline costs."""
    import torch
    N, F, T = 64, 1024, 1024
    Fw, Ng = F // world, N // world
    g = torch.Generator(device="cuda").manual_seed(4242 + rank)
    xe = pkg.clXEngine(1, 2, 0, dev, False, pkg.DTYPE_BYTE, 1, N, 1, 0, Fw, T, [])
    vis = torch.zeros(windows * xe.get_output_buffer_size(), 2, device="cuda")
    ctn = pkg.shard.XEngineCornerTurn(N, F, T, 1, block=xe, windows=windows)
    loc = [torch.randint(-127, 128, ctn.local_shape(), dtype=torch.int8, device="cuda", generator=g) for _ in range(2)]

    def run(k):
        h = ctn.start(loc[0], 0)
        for i in range(k):
            nxt = ctn.start(loc[(i + 1) & 1], (i + 1) & 1) if i + 1 < k else None
            recv = ctn.finish(h)
            xe.xcorrelate_n_device(windows, recv, vis, stations_per_group=Ng)
            h = nxt

    nex = max(3, steps // windows)
    run(2)
    _, ev = time_steps(lambda: run(nex), 1, 0, world)
    ev = max_over_ranks(ev, world)
    dt = ev / (nex * windows)
    # the same batch without the pipeline around it (group-major input already in place)
    # (alternating over both receive slots, as the pipeline does: one slot alone can sit in the Infinity Cache between launches)
    recvs = [ctn.finish(ctn.start(loc[k], k)) for k in range(2)]
    turn = [0]

    def bare_call():
        xe.xcorrelate_n_device(windows, recvs[turn[0] & 1], vis, stations_per_group=Ng)
        turn[0] += 1
    _, evb = time_steps(bare_call, nex, 2, world)
    bare = max_over_ranks(evb, world) / (nex * windows)
    # the exchange alone (pack + all-to-all of `windows` windows, nothing under it): with `bare` this says link-bound or compute-bound directly
    def exchanges():
        for i in range(nex):
            ctn.finish(ctn.start(loc[i & 1], i & 1))
    exchanges()
    _, evx = time_steps(exchanges, 1, 0, world)
    xchg = max_over_ranks(evx, world) / nex
    per_link = int(loc[0].numel() // world)  # bytes a rank sends to ONE peer per exchange (one xGMI link each)
    return {"us_per_integration": round(dt * 1e6, 2), "us_per_integration_bare_batched_call": round(bare * 1e6, 2),
            "pipeline_overhead_us_per_integration": round((dt - bare) * 1e6, 2),
            "alltoall_us_per_exchange": round(xchg * 1e6, 2), "alltoall_us_per_integration": round(xchg * 1e6 / windows, 2),
            "alltoall_bytes_per_link_per_exchange": per_link if world > 1 else 0,
            "alltoall_GBps_per_link": round(per_link / xchg / 1e9, 1) if world > 1 else None,
            "bound": ("exchange" if xchg / windows > bare else "correlation") if world > 1 else "one rank: no exchange",
            "total_input_MSamples_per_s": round(N * F * T / dt / 1e6, 1), "integrations": nex * windows, "windows_per_exchange": windows,
            "n_gpus": world, "channels_per_rank": Fw, "antennas_per_rank_ingest": Ng,
            "alltoall_bytes_sent_per_rank_per_exchange": int(loc[0].numel() * (world - 1) // world), "timing": "HIP events on the launch stream",
            "overlap": "exchange(i+1) on a side stream under correlate(i); receive buffer read in place"}


def sharded_one_process(pkg, dev, windows=8, ranks=2):
    """The single-process driver of the same pipeline (mi355_xengine_shard_*: what a GNU Radio flowgraph can use) with `ranks` logical ranks
    on THIS device: config 5 split into antenna groups, packed, exchanged by (here: device-local) peer copies, correlated per channel slab in
    one launch per rank and exchange, two slots in flight.  One device does all the work of all ranks plus the copies, so the figure is the
    pipeline's overhead against the batched one-GPU call, not a scaling number."""
    import time
    import torch
    N, F, T = 64, 1024, 1024
    sh = pkg.clXEngineSharded([dev] * ranks, 1, N, F, T, windows)
    g = torch.Generator(device="cuda").manual_seed(777)
    frames = [[torch.randint(-127, 128, (sh.frames_bytes(),), dtype=torch.int8, device="cuda", generator=g) for _ in range(ranks)] for _ in range(2)]
    outs = [torch.zeros(windows * sh.slab_items(), 2, device="cuda") for _ in range(ranks)]
    torch.cuda.synchronize()
    for k in range(2):
        sh.submit_device(frames[k & 1], outs)
    sh.synchronize()
    nex = 6
    t0 = time.perf_counter()
    for k in range(nex):
        sh.submit_device(frames[k & 1], outs)
    sh.synchronize()
    dt = (time.perf_counter() - t0) / (nex * windows)
    del frames, outs
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    sh.close()
    return {"us_per_integration": round(dt * 1e6, 2), "logical_ranks_on_this_device": ranks, "windows_per_exchange": windows,
            "timing": "host clock around %d back-to-back submits + synchronize (the handle's own streams)" % nex,
            "device_bytes_per_integration": {"pack_read_write": 2 * T * N * F * 2, "peer_copies_read_write": 2 * T * N * F * 2,
                                             "correlation_in_out": T * N * F * 2 + F * (N * (N + 1) // 2) * 8},
            "note": "one device carries every rank's packing (134 MB read + written per window), the copies that are xGMI transfers on several "
                    "devices (134 MB read + written here) and both ranks' correlations at once: 4.5 x the bytes of the bare call -- a functional run of "
                    "the single-process pipeline inside the bench, not a scaling number"}


def sharded_host_ingest(pkg, dev, windows=4, exchanges=5):
    """The streaming HOST path of the single-process sharded block (mi355_xengine_shard_acquire / submit_acquired / wait: pinned frame slots, every
    rank uploads its antenna group on its own stream, two exchanges in flight -- what clXEngine::work_test uses after set_shard_devices), config 5,
    for 1, 2 and 4 logical ranks on THIS device.  All ranks share the one host link here, so the time per integration cannot drop with the rank
    count; the check is that it does not GROW (serialised uploads or a synchronisation per call would show as W x).  PCIe inclusive, never `value`."""
    import time
    N, F, T = 64, 1024, 1024
    out = {"windows_per_exchange": windows, "exchanges_timed": exchanges, "what": "host clock around acquire + submit_acquired (+ wait once two are in flight) "
           "per exchange, frames already in the pinned slot (the block's gather is not in it); 134 MB up, 17 MB down per integration"}
    for ranks in (1, 2, 4):
        sh = pkg.clXEngineSharded([dev] * ranks, 1, N, F, T, windows)
        res = np.empty(windows * sh.get_output_buffer_size(), np.complex64)
        for _ in range(2):  # both pinned slots, every rank's device buffers
            sh.acquire()[:1] = 1
            sh.submit_acquired()
        while sh.pending():
            sh.wait(res)
        t0 = time.perf_counter()
        for _ in range(exchanges):
            if sh.pending() == 2:
                sh.wait(res)
            sh.acquire()
            sh.submit_acquired()
        while sh.pending():
            sh.wait(res)
        dt = (time.perf_counter() - t0) / (exchanges * windows)
        out["ranks_%d" % ranks] = {"us_per_integration": round(dt * 1e6, 1), "host_GBps_up": round(T * N * F * 2 / dt / 1e9, 1)}
        sh.close()
    out["grows_with_ranks"] = out["ranks_4"]["us_per_integration"] > 1.15 * out["ranks_1"]["us_per_integration"]
    return out


def annotate_sharded_scaling(extras, world):
    """Strong scaling of the 64 x 1024 x 1024 integration stream: efficiency = t(1 GPU) / (N x t(N GPUs)), both sides eight windows per launch
    (like for like), the one-GPU time measured in the SAME line (N > 1: `clXEngine_n1_reference`, every rank's own device; N = 1: the batched
    row, so the figure is the pipeline against the bare batched call and cannot exceed 1 by more than timer noise).  The single call's time is
    carried as a time only: a ratio of a batched run to an unbatched one is not an efficiency (it read 1.4 at N = 1 in round 4).
    Pure bookkeeping on the `blocks` dictionary: tests/test_multi_gpu_cpu.py feeds it made-up times."""
    n1 = extras.get("clXEngine_n1_reference", {}).get("us_per_integration_one_gpu") or extras.get("clXEngine_64ant_1024ch_1024t_ichar", {}).get("us_per_launch")
    row = extras.get("clXEngine_sharded")
    if isinstance(row, dict) and row.get("us_per_integration"):
        if n1:
            row["n1_us_per_integration_single_call"] = n1
        n1b = extras.get("clXEngine_n1_reference", {}).get("us_per_window_one_gpu_8_windows_per_launch") or \
            extras.get("clXEngine_64ant_1024ch_1024t_ichar_batched", {}).get("windows_per_launch_8", {}).get("us_per_window")
        if n1b:
            row["n1_us_per_window_batched"] = n1b
            row["scaling_efficiency_vs_n1_batched"] = round(n1b / (world * row["us_per_integration"]), 3)
    return extras


def baseline_summary(line):
    """The five BASELINE.json configs in ONE compact object (< 2 KB), the LAST key of the line: whatever tail of the line a log keeps, these survive.
    us = HIP-event time per launch; frac = algorithmic bytes / us / 8 TB/s; tx = HBM bytes moved (PMC, profiles/<tag>_baseline_configs.txt) /
    algorithmic bytes; cpu = the oracle's restatement on ONE host core, MSamples/s."""
    b = line.get("blocks") or {}
    try:
        with open(os.path.join(ROOT, "profiles", "baseline_configs_pmc.json")) as fh:
            pmc = json.load(fh)
    except Exception:  # noqa: BLE001
        pmc = {}

    def tx(k):
        return (pmc.get(k) or {}).get("traffic_ratio")

    def row(key, **kw):
        r = b.get(key)
        return r if isinstance(r, dict) else None

    out = {"keys": "us per launch (HIP events), frac of 8 TB/s on algorithmic bytes, tx = PMC bytes / algorithmic (%s), cpu = oracle 1 core MS/s"
                   % pmc.get("source", "no pmc file")}
    c1 = row("config1_clMathOp_testCPU_8192")
    if c1:
        out["1_clMathOp_cmul_8192_testCPU"] = {"cpu_us_per_call": c1.get("us_per_call"), "cpu": c1.get("MSamples_per_s")}
    rf, cb = line.get("roofline") or {}, line.get("cpu_baseline") or {}
    out["2_clFFT_4096_fwd_blackman_shift"] = {"kernel": "k_fft<4096>", "us": rf.get("kernel_us"), "frac": rf.get("frac"), "tx": tx("2"),
                                             "cpu": cb.get("value")}
    r = row("clFilter_fft_65taps")
    if r:
        h = row("clFilter_fft_65taps_32768_hostpath") or {}
        out["3_clFilter_fft_65taps"] = {"kernel": "k_ols<%s>" % r.get("fft_size"), "us": r["us_per_launch"], "frac": r["hbm_frac"], "tx": tx("3"),
                                        "cpu": r.get("cpu_1core_MSamples_per_s"), "host_32768_call_us": h.get("us_per_call_median")}
    r = row("clPolyphaseChannelizer_64x32_stream")
    if r:
        s = row("clPolyphaseChannelizer_64x32_buf65536") or {}
        out["4_clPolyphaseChannelizer_64x32"] = {"kernel": "k_pfbw<64,32>", "us": r["us_per_launch"], "frac": r["hbm_frac"], "tx": tx("4"),
                                                 "cpu": r.get("cpu_1core_MSamples_per_s"), "buf65536_call_us": s.get("us_per_launch"),
                                                 "per_gpu_MSps": r.get("per_gpu_MSamples_per_s")}
    r = row("clXEngine_64ant_1024ch_1024t_ichar")
    if r:
        d = {"kernel": r.get("kernel"), "us": r["us_per_launch"], "frac": r.get("hbm_frac_algorithmic"), "frac_i8_5POPS": r.get("mfma_frac_i8_5POPS"),
             "tx": tx("5"), "cpu": r.get("cpu_1core_MSamples_per_s")}
        w8 = (row("clXEngine_64ant_1024ch_1024t_ichar_batched") or {}).get("windows_per_launch_8")
        if w8:
            d["x8_us_per_window"], d["x8_frac"], d["x8_kernel"], d["x8_tx"] = w8["us_per_window"], w8["hbm_frac_algorithmic"], w8["kernel"], tx("5b")
        out["5_clXEngine_64x1024x1024_ichar"] = d
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary per-block lines")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-sustained", action="store_true", help="skip the >= 2 s back-to-back leg")
    ap.add_argument("--sustain-s", type=float, default=5.0)
    ap.add_argument("--secondary-timeout", type=int, default=300, help="seconds the secondary legs may take before the line is printed without the rest")
    a = ap.parse_args()

    import torch
    rank, world, local = dist_setup(a.gpus)
    pkg = entry.load_package()

    # ---- headline: clFFT 4096 forward, blackman window, shift ---------------------------
    n_k = np.arange(FFT_N)
    window = (0.42 - 0.5 * np.cos(2 * np.pi * n_k / (FFT_N - 1)) + 0.08 * np.cos(4 * np.pi * n_k / (FFT_N - 1))).astype(np.float32)
    g = torch.Generator(device="cuda").manual_seed(1234 + rank)
    x = torch.randn(FRAMES_PER_STEP * FFT_N, 2, device="cuda", generator=g)  # seeded N(0,1) complex fp32
    y = torch.empty_like(x)
    blk = pkg.clFFT(FFT_N, pkg.CLFFT_FORWARD, window, pkg.DTYPE_COMPLEX, 1, 2, 0, local, 0, 1, True)

    def step():
        blk.work_device(FRAMES_PER_STEP, [x], [y])

    samples_per_step = FRAMES_PER_STEP * FFT_N
    # The long leg runs FIRST: the K timed steps that follow then see the clocks of a loaded device instead of the first
    # milliseconds after idle (the same K-step burst measured cold is ~5 % slower than the 2 s median on this device).
    sus = None
    if not a.no_sustained:
        sus = sustained(step, samples_per_step, a.sustain_s)
    wall, ev = time_steps(step, a.steps, a.warmup, world)
    if sus is not None:
        sus = sus()
    wall = max_over_ranks(wall, world)
    ev = max_over_ranks(ev, world)
    value = world * samples_per_step * a.steps / wall / 1e6
    kernel_s = ev / a.steps  # one launch per step: HIP-event time per launch on the launch stream
    achieved = samples_per_step * BYTES_PER_SAMPLE / kernel_s / 1e9

    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", "fft4096_pmc.json")) as fh:
            pmc = json.load(fh)
        traffic = pmc["hbm_bytes_per_launch"]
    except Exception:
        pmc = None
    extras = {}
    line = {
        "metric": "MSamples/sec (complex-float) through clFFT 4096 fwd + window + shift",
        "value": round(value, 1),
        "unit": "MSamples/s",
        "n_gpus": world,
        "steps": a.steps,
        "warmup": a.warmup,
        "ms_per_step": round(wall / a.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": "BASELINE configs[1]: forward clFFT 4096-pt complex, blackman window + fftshift, "
                               "%d frames/step/GPU device-resident (1 GiB in+out)" % FRAMES_PER_STEP,
                   "fft_size": FFT_N, "frames_per_step": FRAMES_PER_STEP, "parallelism": "replica-per-gpu x%d" % world},
        "per_gpu_MSamples_per_s": round(value / world, 1),
        "sustained": sus,
        "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                     "traffic_source": (pmc or {}).get("source"),
                     "kernel": "k_fft<4096,-1,false,1>", "kernel_us": round(kernel_s * 1e6, 2),
                     "algorithmic_bytes_per_launch": samples_per_step * BYTES_PER_SAMPLE},
        "cpu_baseline": None,
        "blocks": extras,
    }
    # The headline is measured; everything below is secondary.  A watchdog guarantees the ONE JSON line even if a secondary
    # leg hangs (e.g. a rank lost inside a collective of the sharded X-engine): rank 0 prints what exists, every rank exits.
    import threading
    emitted = threading.Event()

    def emit(note=None):
        if emitted.is_set():
            return
        emitted.set()
        if note:
            line["watchdog"] = note
        try:
            line["baseline_configs"] = baseline_summary(line)  # the last key of the line
        except Exception as exc:  # noqa: BLE001
            line["baseline_configs"] = {"error": "%s: %s" % (type(exc).__name__, exc)}
        if rank == 0:
            print(json.dumps(line), flush=True)

    def fire():
        emit("secondary legs exceeded %d s: line printed with what was measured" % a.secondary_timeout)
        os._exit(0)

    dog = threading.Timer(a.secondary_timeout, fire)
    dog.daemon = True
    dog.start()
    taps_pair = (lowpass_taps(1.0, 10e6, 1e6, 372000.0), np.concatenate([lowpass_taps(1.0, 64.0, 0.5, 0.0753), [0.0]]).astype(np.float32))
    if not a.no_extra:
        # fixture taps: firdes.low_pass(1,10e6,1e6,372e3) = 65 taps; low_pass(1,64,.5,.0753)+[0] = 2048 taps.
        # Designed by the product-independent formula above (same definition as tests/golden/gen_golden.py).
        # The secondary lines must never cost the headline: a failure here is reported in the line, not raised.
        try:
            extra_blocks(pkg, taps_pair, local, max(5, a.steps // 5), 2, world, rank, extras)
        except Exception as exc:  # noqa: BLE001
            extras["error"] = "%s: %s" % (type(exc).__name__, exc)
        # the only data-path collective of the hot path, INSIDE the JSON line (world 1: the same pipeline, no peers)
        try:
            extras["clXEngine_sharded"] = sharded_xengine(pkg, local, max(20, a.steps // 2), world, rank)
            annotate_sharded_scaling(extras, world)
        except Exception as exc:  # noqa: BLE001
            extras["clXEngine_sharded"] = {"error": "%s: %s" % (type(exc).__name__, exc)}
        if rank == 0 and world == 1:
            try:
                extras["clXEngine_shard_one_process"] = sharded_one_process(pkg, local)
            except Exception as exc:  # noqa: BLE001
                extras["clXEngine_shard_one_process"] = {"error": "%s: %s" % (type(exc).__name__, exc)}
            try:
                extras["clXEngine_shard_host_ingest"] = sharded_host_ingest(pkg, local)
            except Exception as exc:  # noqa: BLE001
                extras["clXEngine_shard_host_ingest"] = {"error": "%s: %s" % (type(exc).__name__, exc)}
            try:
                extras.update(hostpath_blocks(pkg, taps_pair, local))
            except Exception as exc:  # noqa: BLE001
                extras["hostpath_error"] = "%s: %s" % (type(exc).__name__, exc)
    if rank == 0 and world == 1 and not a.no_cpu:
        line["cpu_baseline"] = cpu_baseline_fft(entry.load_oracle(), window)
        # BASELINE configs[0] and the secondary blocks' one-core legs are timed from C (oracle/cpu_bench); the same calls made from Python
        # through ctypes stay beside them (an 8192-item call is dominated by the interpreter there; SURVEY 8d asks for the CLI's loop)
        c1 = config1_testcpu(entry.load_oracle())
        harness = cpu_harness(1.0)
        if harness:
            h1 = dict(harness["config1_clMathOp_testCPU_8192"])
            h1["kind"] = "port (oracle mathop, the CLI's loop in C: 1 warm-up + 200 timed calls, one core)"
            h1["through_ctypes_us_per_call"] = c1["us_per_call"]
            h1["through_ctypes_MSamples_per_s"] = c1["MSamples_per_s"]
            extras["config1_clMathOp_testCPU_8192"] = h1
        else:
            extras["config1_clMathOp_testCPU_8192"] = c1
        if not a.no_extra:
            for k, v in cpu_extras(entry.load_oracle(), taps_pair).items():
                if k in extras and isinstance(extras[k], dict):
                    if isinstance(v, dict):
                        extras[k]["cpu_1core_MSamples_per_s"] = v["MSamples_per_s"]
                        extras[k]["cpu_1core_sample"] = v["sample"]
                    else:
                        extras[k]["cpu_1core_MSamples_per_s"] = v
            if harness:  # the C-timed figures replace the ctypes ones, which keep a key of their own
                names = {"clComplexFilter_fft_65ctaps": "clComplexFilter_fir_65ctaps"}
                for k in list(extras):
                    hk = names.get(k, k)
                    if isinstance(extras[k], dict) and "cpu_1core_MSamples_per_s" in extras[k] and hk in harness:
                        hv = harness[hk]
                        extras[k]["cpu_1core_through_ctypes_MSamples_per_s"] = extras[k]["cpu_1core_MSamples_per_s"]
                        extras[k]["cpu_1core_MSamples_per_s"] = hv["MSamples_per_s"] if isinstance(hv, dict) else hv
                extras["cpu_harness"] = {"kind": harness["kind"], "seconds_per_leg": harness["seconds_per_leg"],
                                         "clFFT_4096_blackman_shift_MSamples_per_s": harness["clFFT_4096_blackman_shift"]}
    dog.cancel()
    emit()
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
