#!/usr/bin/env python3
"""Random clFFT shapes against numpy's float64 pocketfft with clFFT_impl's window / shift semantics (oracle/o_fft.c:140-188; the numpy form is tied to the
oracle in tests/test_fft_gpu.py): lengths 2 ... 20000 (powers of two, 2^a 3^b 5^c 7^d 11^e 13^f, anything else = chirp-z), both directions, window on / off,
shift on / off, complex and real input, 1 ... 3000 frames (ragged against the workgroups' frame counts).
usage: python tools/stress/fft_fuzz.py [seconds] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import __graft_entry__ as e
pkg = e.load_package()
ARGS = (1, 2, 0, 0)
T_END = time.time() + (float(sys.argv[1]) if len(sys.argv) > 1 else 60)
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)


def smooth(limit):
    n = 1
    while True:
        f = int(rng.choice([2, 2, 2, 3, 3, 5, 5, 7, 11, 13]))
        if n * f > limit:
            return max(n, 2)
        n *= f
        if rng.random() < 0.25 and n >= 6:
            return n


def ref_block(n, fwd, w, shift, x):
    x = x.astype(np.complex128).reshape(-1, n)
    if w is not None:
        x = x * np.asarray(w, np.float64)
    if not fwd and shift:
        half = n // 2
        x = np.concatenate([x[:, half:], x[:, :half]], axis=1)
    y = np.fft.fft(x, axis=1) if fwd else np.fft.ifft(x, axis=1) * n
    if fwd and shift:
        ln = (n + 1) // 2
        y = np.concatenate([y[:, ln:], y[:, :ln]], axis=1)
    return y.reshape(-1)


cases = bad = 0
kinds = {"pow2": 0, "mixed": 0, "other": 0}
while time.time() < T_END:
    r = rng.random()
    if r < 0.25:
        n = 1 << int(rng.integers(1, 15)); kind = "pow2"
    elif r < 0.85:
        n = smooth(int(rng.choice([64, 256, 1024, 4096, 20000]))); kind = "pow2" if n & (n - 1) == 0 else "mixed"
    else:
        n = int(rng.integers(3, 5000)); kind = "other"
    fwd, shift, win, real = bool(rng.integers(2)), bool(rng.integers(2)), bool(rng.integers(2)), rng.random() < 0.25
    nvec = int(rng.choice([1, 2, 5, 17, 64, 300, 1000, 3000]))
    if nvec * n > 1 << 22:
        nvec = max(1, (1 << 22) // n)
    w = (0.5 + rng.random(n)).astype(np.float32) if win else None
    try:
        blk = pkg.clFFT(n, pkg.CLFFT_FORWARD if fwd else pkg.CLFFT_BACKWARD, [] if w is None else w, pkg.DTYPE_FLOAT if real else pkg.DTYPE_COMPLEX, *ARGS, 0, 1, shift)
    except Exception as exc:  # noqa: BLE001
        print("create failed n=%d" % n, exc, flush=True)
        bad += 1
        continue
    if real:
        x = rng.standard_normal(nvec * n).astype(np.float32)
        xd = torch.from_numpy(x).cuda()
        xr = x
    else:
        x = rng.standard_normal((nvec * n, 2)).astype(np.float32)
        xd = torch.from_numpy(x).cuda()
        xr = x.view(np.complex64).reshape(-1)
    yd = torch.full((nvec * n, 2), 7.0, device="cuda")
    blk.work_device(nvec, [xd], [yd])
    torch.cuda.synchronize()
    y = yd.cpu().numpy().view(np.complex64).reshape(-1)
    ref = ref_block(n, fwd, w, shift, xr)
    err = np.abs(y - ref).max() / max(np.abs(ref).max(), 1e-30)
    cases += 1
    kinds[kind] += 1
    if not np.isfinite(err) or err > 2e-5:
        bad += 1
        print("MISMATCH n=%d fwd=%d shift=%d win=%d real=%d nvec=%d err %.3g" % (n, fwd, shift, win, real, nvec, err), flush=True)
    del blk
print("fft fuzz: %d cases, %d bad; %s" % (cases, bad, kinds), flush=True)
