"""GPU probe: headline clFFT (4096 fwd, window + shift, 16384 frames) sustained time per launch under env-var variants."""
import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, time
sys.path.insert(0, %r)
import numpy as np, torch
import __graft_entry__ as e
pkg = e.load_package()
N, FR = 4096, 16384
w = np.blackman(N).astype(np.float32)
x = torch.randn(FR * N, 2, device="cuda"); y = torch.empty_like(x)
blk = pkg.clFFT(N, pkg.CLFFT_FORWARD, w, pkg.DTYPE_COMPLEX, 1, 2, 0, 0, 0, 1, True)
f = lambda: blk.work_device(FR, [x], [y])
for _ in range(200): f()
torch.cuda.synchronize()
per = []
for _ in range(12):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(100): f()
    b.record(); torch.cuda.synchronize()
    per.append(a.elapsed_time(b) * 10)
per.sort()
print("%%.2f %%.2f" %% (per[len(per)//2], per[0]))
''' % ROOT
variants = [{}, {"MI355_FFT_WG_PER_CU": "16"}, {"MI355_FFT_WG_PER_CU": "20"}, {"MI355_FFT_WG_PER_CU": "24"}, {"MI355_FFT_WG_PER_CU": "28"}] + \
           [{"MI355_FFT_PREFETCH": "1", "MI355_FFT_WG_PER_CU": str(k)} for k in (1, 2, 3, 5, 6, 10, 12)] + [{}, {"MI355_FFT_WG_PER_CU": "16"}]
if len(sys.argv) > 1:
    variants = json.loads(sys.argv[1])
for v in variants:
    env = dict(os.environ, **v)
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
    print(v, r.stdout.strip() or r.stderr[-300:], flush=True)
