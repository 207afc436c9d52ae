"""X-engine zero-copy host pipeline (acquire / gather / submit_acquired / wait), time of every call (tuning aid)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as e
pkg = e.load_package()
N, F, T, PER = 64, 1024, 1024, 256
xe = pkg.clXEngine(1, 2, 0, 0, False, pkg.DTYPE_BYTE, 1, N, 1, 0, F, T, [])
L, h = xe._L, xe._h
frames = np.full(PER * F * 2, 3, np.int8)
ins = (C.c_void_p * N)(*([frames.ctypes.data] * N))
vis = np.empty(xe.get_output_buffer_size(), np.complex64)
vp = vis.ctypes.data_as(C.c_void_p)
names = ["acquire", "gather0", "gather1", "gather2", "gather3", "submit", "wait"]
acc = {k: [] for k in names}
def window(collect):
    fb = C.c_void_p()
    t = [time.perf_counter()]
    assert L.mi355_xengine_acquire(h, C.byref(fb)) == 0; t.append(time.perf_counter())
    for q in range(T // PER):
        assert L.mi355_xengine_gather(h, PER, q * PER, ins, fb) == 0; t.append(time.perf_counter())
    assert L.mi355_xengine_submit_acquired(h, None) == 0; t.append(time.perf_counter())
    if collect:
        assert L.mi355_xengine_wait(h, vp) == 0
    t.append(time.perf_counter())
    return [b - a for a, b in zip(t, t[1:])]
window(False); window(True); window(True)
t0 = time.perf_counter()
K = 8
for _ in range(K):
    for k, v in zip(names, window(True)): acc[k].append(v)
dt = (time.perf_counter() - t0) / K
L.mi355_xengine_wait(h, vp)
print("per window %.2f ms" % (dt * 1e3))
for k in names: print("  %-8s mean %.3f ms  min %.3f  max %.3f" % (k, np.mean(acc[k]) * 1e3, min(acc[k]) * 1e3, max(acc[k]) * 1e3))
