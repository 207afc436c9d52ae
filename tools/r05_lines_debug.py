"""debug aid: which (channel, row-tile pair) of the whole-line kernel's output differs from the oracle"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MI355_XE_LINES_MIN_UNITS", "4")
import numpy as np, torch
import __graft_entry__ as e
pkg = e.load_package()
N, F, T, nint = 64, int(os.environ.get("F", 64)), int(os.environ.get("T", 32)), int(os.environ.get("NINT", 1))
rng = np.random.default_rng(1)
w = rng.integers(-128, 128, size=(nint, T, N, F, 1, 2), dtype=np.int64).astype(np.int8)
xe = pkg.clXEngine(1, 2, 0, 0, False, pkg.DTYPE_BYTE, 1, N, 1, 0, F, T, [])
per = xe.get_output_buffer_size()
out = torch.zeros(nint * per, 2, device="cuda")
xe.xcorrelate_n_device(nint, torch.from_numpy(w).cuda(), out)
torch.cuda.synchronize()
got = out.cpu().numpy().view(np.complex64).reshape(nint, F, -1)
o = e.load_oracle(); o.lib()
ref = np.stack([o.xengine_ichar(N, F, 1, T, w[i].reshape(-1), exact=True).reshape(F, -1) for i in range(nint)])
bad = got != ref
print("mismatches", bad.sum(), "of", bad.size)
tri = [(a, b) for a in range(N) for b in range(a + 1)]
tile = np.array([(a // 16) * 4 + b // 16 for a, b in tri])
for p in sorted(set(tile)):
    sel = tile == p
    print("pair %d%d: bad %d of %d  (re bad %d, im bad %d)" % (p // 4, p % 4, bad[..., sel].sum(), bad[..., sel].size,
          (got.real[..., sel] != ref.real[..., sel]).sum(), (got.imag[..., sel] != ref.imag[..., sel]).sum()))
badch = bad.any(axis=(0, 2))
print("bad channels:", np.nonzero(badch)[0][:64])
i = np.argwhere(bad)
if len(i):
    k = tuple(i[0]); print("first:", k, tri[k[2]], got[k], ref[k], got[k] / (1 / 127.0) ** 2, ref[k] / (1 / 127.0) ** 2)
