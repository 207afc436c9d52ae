"""clPolyphaseChannelizer 64 x 32 stream kernel: phase removal A/B inside one process (round 5 tuning aid; MI355_PFB_DBG bits: 1 two taps
instead of 32, 2 no DFT, 4 no stores -- wrong results, same access pattern)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as e
pkg = e.load_package(); o = e.load_oracle()
N = 1 << 26
t = np.concatenate([o.firdes_low_pass(1.0, 64.0, 0.5, 0.0753), [0.0]]).astype(np.float32)
p = pkg.clPolyphaseChannelizer(1, 2, 0, 0, t, N, 64, 64, list(range(64)))
x = torch.randn(N + 2048 - 64, 2, device="cuda"); y = torch.empty(N, 2, device="cuda")
def ev(fn, it=int(os.environ.get('PROBE_IT', '400'))):
    for _ in range(100): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e-3 / it
for arm in sys.argv[1:] or ["base"]:
    sets = [] if arm == "base" else [kv.split("=") for kv in arm.split(",")]
    for k, v in sets: os.environ[k] = v
    dt = ev(lambda: p.work_device([x], [y]))
    print("%-40s %.1f us %.1f GS/s %.3f of 8 TB/s" % (arm, dt * 1e6, N / dt / 1e9, N * 16 / dt / 8e12), flush=True)
    for k, v in sets: del os.environ[k]
