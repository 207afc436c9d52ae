"""clXEngine config 5 (64 antennas x 1024 channels x 1024 frames, IChar): windows per launch x time split (MI355_XE_TSPLIT) -- tuning aid."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as e
pkg = e.load_package()
N, F, T = 64, int(os.environ.get("PROBE_F", "1024")), 1024
nint = int(os.environ.get("PROBE_NINT", "8"))
it = int(os.environ.get("PROBE_IT", "50"))
xe = pkg.clXEngine(1, 2, 0, 0, False, pkg.DTYPE_BYTE, 1, N, 1, 0, F, T, [])
x = torch.randint(-127, 128, (nint, T, N, F, 1, 2), dtype=torch.int8, device="cuda")
out = torch.zeros(nint * xe.get_output_buffer_size(), 2, device="cuda")
fn = (lambda: xe.xcorrelate_n_device(nint, x, out)) if nint > 1 else (lambda: xe.xcorrelate_device(x, out))
for _ in range(5): fn()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(it): fn()
b.record(); torch.cuda.synchronize()
dt = a.elapsed_time(b) * 1e-3 / it
print("xe F=%d nint=%d tsplit=%s: %.1f us per launch, %.2f us per window" % (F, nint, os.environ.get("MI355_XE_TSPLIT", "auto"), dt * 1e6, dt * 1e6 / nint))
