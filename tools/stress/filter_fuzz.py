#!/usr/bin/env python3
"""Random clFilter / clComplexFilter shapes against a float64 convolution: 1 ... 3000 taps (real and complex), decimations 1 ... 40 (odd and even), FFT mode
and time-domain mode, 1 ... 2^18 outputs.  usage: python tools/stress/filter_fuzz.py [seconds] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from scipy.signal import fftconvolve
import __graft_entry__ as e
pkg = e.load_package()
ARGS = (1, 2, 0, 0)
T_END = time.time() + (float(sys.argv[1]) if len(sys.argv) > 1 else 60)
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
cases = bad = 0
kinds = {}
while time.time() < T_END:
    nt = int(rng.choice([1, 2, 3, 7, 15, 16, 17, 33, 64, 65, 66, 96, 129, 255, 300, 497, 498, 600, 1000, 2049, 3000]))
    if rng.random() < 0.3:
        nt = int(rng.integers(1, 700))
    dec = int(rng.choice([1, 1, 2, 3, 4, 5, 6, 7, 8, 9, 12, 16, 17, 25, 32, 40]))
    use_time = bool(rng.integers(2))
    ctaps = rng.random() < 0.3
    if use_time and nt > 700:
        nt = int(rng.integers(400, 700))
    nout = int(rng.choice([1, 5, 64, 1000, 4099, 65536, 100003, 262144]))
    if nout * dec > 1 << 22:
        nout = max(1, (1 << 22) // dec)
    taps = (rng.standard_normal(nt) / np.sqrt(nt)).astype(np.float32)
    if ctaps:
        taps = (taps + 1j * (rng.standard_normal(nt) / np.sqrt(nt))).astype(np.complex64)
    nin = nout * dec + nt - 1  # (GNU Radio: noutput_items x decimation + history)
    x = rng.standard_normal((nin, 2)).astype(np.float32)
    try:
        blk = pkg.clComplexFilter(*ARGS, dec, taps, 1, 0, use_time=use_time) if ctaps else pkg.clFilter(*ARGS, dec, taps, 1, 0, use_time)
    except Exception as exc:  # noqa: BLE001
        print("create failed", nt, dec, use_time, ctaps, exc, flush=True)
        bad += 1
        continue
    xd = torch.from_numpy(x).cuda()
    yd = torch.full((nout + 64, 2), 7.0, device="cuda")
    blk.work_device(nout, [xd], [yd[:nout]])
    torch.cuda.synchronize()
    y = yd[:nout].cpu().numpy().view(np.complex64).reshape(-1)
    xc = x.view(np.complex64).reshape(-1).astype(np.complex128)
    full = fftconvolve(xc, taps.astype(np.complex128), mode="valid") if nt > 64 else np.convolve(xc, taps.astype(np.complex128), mode="valid")
    ref = full[::dec][:nout]
    err = np.abs(y - ref).max() / max(np.abs(ref).max(), 1e-30)
    guard = bool(torch.all(yd[nout:] == 7.0))
    cases += 1
    k = ("time" if use_time else "fft", "ctaps" if ctaps else "rtaps", "dec>8" if dec > 8 else "dec<=8")
    kinds[k] = kinds.get(k, 0) + 1
    if not np.isfinite(err) or err > 2e-5 or not guard:
        bad += 1
        print("MISMATCH taps=%d complex=%d dec=%d time=%d nout=%d err %.3g guard %s" % (nt, ctaps, dec, use_time, nout, err, guard), flush=True)
    del blk
print("filter fuzz: %d cases, %d bad; %s" % (cases, bad, kinds), flush=True)
