#!/usr/bin/env python3
"""Random channelizer shapes against the oracle (float64): channel counts 2 ... 1100 (powers of two, mixed-radix, primes), 1 ... 70 taps per arm with ragged
last arms, critically sampled and oversampled, whole / partial / permuted channel maps, 1 ... 400 steps, single and batched device calls.
usage: python tools/stress/pfb_fuzz.py [seconds] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import __graft_entry__ as e
pkg = e.load_package()
orc = e.load_oracle()  # (the checker: tools/stress is test infrastructure, like tests/)
orc.lib()
ARGS = (1, 2, 0, 0)
T_END = time.time() + (float(sys.argv[1]) if len(sys.argv) > 1 else 60)
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
POOL = [2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 6, 9, 10, 12, 14, 15, 18, 20, 24, 30, 36, 48, 60, 96, 100, 120, 200, 250, 360, 500, 504, 510, 600, 768, 1000, 1100,
        3, 5, 7, 11, 13, 17, 211]
n = bad = 0
kinds = {}
while time.time() < T_END:
    M = int(rng.choice(POOL))
    P = int(rng.choice([1, 2, 3, 5, 8, 9, 16, 17, 31, 32, 33, 48, 64, 70]))
    if M * P > 70000:
        P = max(1, 70000 // M)
    K = M * P - (int(rng.integers(0, M)) if P > 1 and rng.random() < 0.5 else 0)
    R = M
    if rng.random() < 0.2:
        divs = [d for d in (2, 4, 3, 5) if M % d == 0]
        if divs:
            R = M // int(rng.choice(divs))
    steps = int(rng.choice([1, 7, 16, 33, 100, 257, 400]))
    if steps * M > 200000:
        steps = max(1, 200000 // M)
    buf = steps * R
    while buf % M:
        steps += 1
        buf = steps * R
    mode = rng.random()
    chmap = list(range(M)) if mode < 0.6 else [int(v) for v in rng.permutation(M)[:max(1, int(rng.integers(1, M + 1)))]]
    taps = (rng.standard_normal(K) / np.sqrt(P)).astype(np.float32)
    k = int(rng.choice([1, 1, 3]))
    try:
        blk = pkg.clPolyphaseChannelizer(*ARGS, taps, buf, M, R, chmap)
    except Exception as exc:  # noqa: BLE001
        print("create failed", M, P, K, R, steps, len(chmap), exc, flush=True)
        bad += 1
        continue
    x = (rng.standard_normal((k * buf - R + K, 2))).astype(np.float32)
    xd = torch.from_numpy(x).cuda()
    yd = torch.full((k * blk.noutput(), 2), 7.0, device="cuda")
    blk.work_device([xd], [yd], nbuf=k) if k > 1 else blk.work_device([xd], [yd])
    torch.cuda.synchronize()
    y = yd.cpu().numpy().view(np.complex64).reshape(-1)
    xc = x.view(np.complex64).reshape(-1)
    ok = True
    for b in range(k):
        ref = orc.pfb(taps, buf, M, R, chmap, xc[b * buf:b * buf + blk.ninput()], f64=True)
        got = y[b * blk.noutput():(b + 1) * blk.noutput()]
        err = np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-30)
        if not np.isfinite(err) or err > 2e-5:
            ok = False
            print("MISMATCH M=%d P=%d K=%d R=%d steps=%d nmap=%d k=%d buffer %d err %.3g" % (M, P, K, R, steps, len(chmap), k, b, err), flush=True)
            break
    n += 1
    bad += not ok
    kinds[(M & (M - 1) == 0, R == M, P <= 32)] = kinds.get((M & (M - 1) == 0, R == M, P <= 32), 0) + 1
    del blk
print("pfb fuzz: %d cases, %d bad; (power of two, critically sampled, <= 32 taps per arm) -> cases: %s" % (n, bad, kinds), flush=True)
