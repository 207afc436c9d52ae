"""clPolyphaseChannelizer over unusual geometries (channel counts that are not a power of two, oversampling, partial channel maps): a hunt
for rate cliffs (tuning aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as e
pkg = e.load_package()
N = 1 << 26
a = torch.randn(N, 2, device="cuda"); c = torch.empty_like(a)
def ev(fn, it=5):
    for _ in range(2): fn()
    s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    t.record(); torch.cuda.synchronize()
    return s.elapsed_time(t) * 1e-3 / it
rng = np.random.default_rng(1)
for M, R, nmap in [(3,3,3),(5,5,5),(10,10,10),(12,12,12),(20,20,20),(48,48,48),(96,96,96),(100,100,100),(200,200,200),(512,512,512),(1024,1024,1024),(64,32,64),(64,16,64),(100,50,100),(100,25,100),(60,20,60),(256,64,256),(64,64,64),(64,64,16),(64,64,3),(256,256,32),(100,100,10),(1024,1024,100)]:
    for tpa in (8, 32, 48):
        try:
            tp = rng.standard_normal(M * tpa).astype(np.float32)
            steps = ((N // 2) // max(M, R)) // 4 * 4
            buf = steps * R  # input items per call; the call produces steps x nmap outputs
            blk = pkg.clPolyphaseChannelizer(1, 2, 0, 0, tp, buf, M, R, list(range(nmap)))
            dt = ev(lambda: blk.work_device([a], [c]))
            nin = steps * R
            print("PFB M=%4d R=%4d nmap=%4d taps/arm=%2d: %8.1f us  in %6.1f GS/s  (in+out %5.2f TB/s)" % (M, R, nmap, tpa, dt * 1e6, nin / dt / 1e9, (nin + steps * nmap) * 8 / dt / 1e12), flush=True)
        except Exception as ex:
            print("PFB M=%d R=%d taps/arm=%d: %s" % (M, R, tpa, str(ex)[:100]), flush=True)
