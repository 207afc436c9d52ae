import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as e
pkg = e.load_package(); o = e.load_oracle()
N, F, T = 64, 128, 512
nb = N * (N + 1) // 2
def check(x, tag):
    xe = pkg.clXEngine(1, 2, 0, 0, False, pkg.DTYPE_BYTE, 1, N, 1, 0, F, T, [])
    out = np.zeros(xe.get_output_buffer_size(), np.complex64)
    ref = o.xengine_ichar(N, F, 1, T, x, exact=True)
    xe.xcorrelate(x, out)
    d = np.rint((out - ref) * 127 * 127)
    bad = np.nonzero(d != 0)[0]
    print(tag, "mismatches", bad.size)
    if bad.size:
        f = bad // nb; k = bad % nb
        s1 = np.floor(-0.5 + np.sqrt(0.25 + 2 * k)).astype(int); s2 = k - s1 * (s1 + 1) // 2
        import collections
        cnt = collections.Counter(zip((s1 // 16).tolist(), (s2 // 16).tolist()))
        print("  by tile pair:", sorted(cnt.items()))
        cnt = collections.Counter((s1 % 16).tolist()); print("  by row in tile (s1%16):", sorted(cnt.items()))
        cnt = collections.Counter((s2 % 16).tolist()); print("  by col in tile (s2%16):", sorted(cnt.items()))
        cnt = collections.Counter((f % 16).tolist()); print("  by channel in slice:", sorted(cnt.items()))
        cnt = collections.Counter((f // 16).tolist()); print("  by slice:", sorted(cnt.items()))
        vals = collections.Counter(d[bad].tolist()); print("  diff values:", vals.most_common(6))
x = np.zeros((T, N, F, 2), np.int8); x[..., 0] = 1
check(x.reshape(-1), "const (1,0)")
x = np.zeros((T, N, F, 2), np.int8); x[..., 0] = 100; x[..., 1] = -50
check(x.reshape(-1), "const (100,-50)")
rng = np.random.default_rng(1)
check(rng.integers(-128, 128, size=T * N * F * 2, dtype=np.int64).astype(np.int8), "random")
