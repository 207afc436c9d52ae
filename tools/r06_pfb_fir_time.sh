# two-kernel channelizer shapes: us per 2^25 items with the streaming branch-filter kernel (k_pfb_fir) and with k_pfb_branches_t
for v in 0 1; do echo -n "NO_FIR_RING=$v: "; if [ $v = 1 ]; then export MI355_PFB_NO_FIR_RING=1; fi
python - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import __graft_entry__ as e
pkg = e.load_package()
n = 1 << 26
a = torch.randn(n, 2, device="cuda"); c = torch.empty_like(a)
rng = np.random.default_rng(1)
out = []
for M, P in [(1024, 32), (1024, 8), (1024, 16), (2048, 16), (1000, 32), (64, 64), (100, 48), (4096, 4)]:
    taps = rng.standard_normal(M * P).astype(np.float32)
    buf = ((n // 2) // M) * M
    blk = pkg.clPolyphaseChannelizer(1, 2, 0, 0, taps, buf, M, M, list(range(M)))
    xi, yo = a[:blk.ninput()], c[:blk.noutput()]
    for _ in range(3): blk.work_device([xi], [yo])
    torch.cuda.synchronize()
    s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): blk.work_device([xi], [yo])
    t.record(); torch.cuda.synchronize()
    out.append("%dx%d %.0f" % (M, P, s.elapsed_time(t) * 100))
print(" ".join(out))
PY
done
