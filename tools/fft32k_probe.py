"""clFFT 32768 forward, window + shift, 2^26 samples: rate + correctness against torch.fft (tuning aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as e
pkg = e.load_package()
n, N = int(os.environ.get("PROBE_N", "32768")), 1 << 26
w = np.blackman(n).astype(np.float32)
blk = pkg.clFFT(n, pkg.CLFFT_FORWARD, w, pkg.DTYPE_COMPLEX, 1, 2, 0, 0, 0, 1, True)
x = torch.randn(N, 2, device="cuda"); y = torch.empty_like(x)
nv = N // n
fn = lambda: blk.work_device(nv, [x], [y])
for _ in range(5): fn()
it = int(os.environ.get("PROBE_IT", "200"))
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(it): fn()
b.record(); torch.cuda.synchronize()
dt = a.elapsed_time(b) * 1e-3 / it
xc = torch.view_as_complex(x[: 4 * n].reshape(4, n, 2).contiguous()).to(torch.complex128) * torch.from_numpy(w).cuda().to(torch.float64)
ref = torch.fft.fftshift(torch.fft.fft(xc, dim=1), dim=1)
got = torch.view_as_complex(y[: 4 * n].reshape(4, n, 2).contiguous()).to(torch.complex128)
err = float((got - ref).abs().max() / ref.abs().max())
print("clFFT %d: %.1f us per 2^26 samples  %.3f of 8 TB/s  relerr %.2e" % (n, dt * 1e6, N * 16 / dt / 8e12, err))
