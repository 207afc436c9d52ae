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
time.  `roofline` prices the FFT kernel against HBM (16 algorithmic bytes per complex
sample); `cpu_baseline` times the oracle's restatement of the reference's
CPU path (clFFT_impl::testCPU) on one host core over a bounded sample.
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
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    else:
        torch.cuda.set_device(local)
    if world != ngpus:
        raise SystemExit("bench.py --gpus %d but WORLD_SIZE=%d: launch N>1 through torch.distributed.run" % (ngpus, world))
    return rank, world, local


def barrier(world):
    import torch
    torch.cuda.synchronize()
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
    torch.cuda.synchronize()


def time_steps(fn, steps, warmup, world):
    """W untimed steps, then exactly K steps bracketed by barrier+synchronize; returns
    (wall seconds for K steps on this rank, HIP-event seconds for the same K launches)."""
    import torch
    for _ in range(warmup):
        fn()
    barrier(world)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(steps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    barrier(world)
    return wall, e0.elapsed_time(e1) * 1e-3


def max_over_ranks(x, world):
    if world == 1:
        return x
    import torch
    import torch.distributed as dist
    t = torch.tensor([x], dtype=torch.float64, device="cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def cpu_baseline_fft(o, window, budget_s=12.0):
    """Oracle restatement of clFFT_impl::testCPU (window, float FFT, shift) on ONE core."""
    rng = np.random.default_rng(1234)
    probe = 64
    x = (rng.standard_normal(probe * FFT_N) + 1j * rng.standard_normal(probe * FFT_N)).astype(np.complex64)
    t0 = time.perf_counter()
    o.fft_block(FFT_N, True, window, True, o.DTYPE_COMPLEX, x)
    per_frame = (time.perf_counter() - t0) / probe
    frames = int(max(probe, min(200000, budget_s / per_frame)))
    reps = max(1, frames // probe)
    t0 = time.perf_counter()
    for _ in range(reps):
        o.fft_block(FFT_N, True, window, True, o.DTYPE_COMPLEX, x)
    dt = time.perf_counter() - t0
    return {"value": round(reps * probe * FFT_N / dt / 1e6, 2), "unit": "MSamples/s", "cores": 1, "kind": "port",
            "sample": "%d frames of %d-pt complex FFT (window+shift), oracle fft_block f32, %.1f s" % (reps * probe, FFT_N, dt)}


def extra_blocks(pkg, dev, steps, warmup, world):
    """Secondary lines: the other hot-path blocks, device resident, same timing method."""
    import torch
    out = {}
    args = (1, 2, 0, dev)

    def rate(fn, nsamples, bytes_per_sample):
        _, ev = time_steps(fn, steps, warmup, world)
        dt = ev / steps
        return {"MSamples_per_s": round(nsamples / dt / 1e6, 1), "GBps": round(nsamples * bytes_per_sample / dt / 1e9, 1),
                "hbm_frac": round(nsamples * bytes_per_sample / dt / 1e9 / HBM_PEAK_GBS, 4)}

    n = 1 << 25
    a = torch.randn(n, 2, device="cuda")
    b = torch.randn(n, 2, device="cuda")
    c = torch.empty_like(a)
    mul = pkg.clMathOp(pkg.DTYPE_COMPLEX, *args, pkg.MATHOP_MULTIPLY)
    out["clMathOp_multiply_complex"] = rate(lambda: mul.work_device(n, [a, b], [c]), n, 24)
    mc = pkg.clMathConst(pkg.DTYPE_COMPLEX, *args, 2.0, pkg.MATHOP_MULTIPLY)
    out["clMathConst_multiply_complex"] = rate(lambda: mc.work_device(n, [a], [c]), n, 16)
    del a, b, c
    torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary per-block lines")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
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

    wall, ev = time_steps(step, a.steps, a.warmup, world)
    wall = max_over_ranks(wall, world)
    ev = max_over_ranks(ev, world)
    samples_per_step = FRAMES_PER_STEP * FFT_N
    value = world * samples_per_step * a.steps / wall / 1e6
    kernel_s = ev / a.steps  # one launch per step: HIP-event time per launch on the launch stream
    achieved = samples_per_step * BYTES_PER_SAMPLE / kernel_s / 1e9

    extras = {}
    if not a.no_extra:
        extras = extra_blocks(pkg, local, max(5, a.steps // 5), 2, world)
    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu:
        cpu = cpu_baseline_fft(entry.load_oracle(), window)

    if rank == 0:
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
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
                         "kernel": "k_fft<4096,-1,false>", "kernel_us": round(kernel_s * 1e6, 2),
                         "algorithmic_bytes_per_launch": samples_per_step * BYTES_PER_SAMPLE},
            "cpu_baseline": cpu,
            "blocks": extras,
        }
        print(json.dumps(line))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
