"""clMathOp / clMathConst / elementwise family over data types, operations and odd call sizes (tuning aid: looks for rate cliffs)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as e
pkg = e.load_package()
def ev(fn, it=10):
    for _ in range(2): fn()
    s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    t.record(); torch.cuda.synchronize()
    return s.elapsed_time(t) * 1e-3 / it
N = (1 << 26) + 7
a = torch.randn(N, 2, device="cuda"); b = torch.randn(N, 2, device="cuda"); c = torch.empty_like(a)
ops = [k for k in dir(pkg) if k.startswith("MATHOP_")]
for dt, name, w in ((pkg.DTYPE_COMPLEX, "complex", 8), (pkg.DTYPE_FLOAT, "float", 4), (pkg.DTYPE_INT, "int", 4)):
    for op in ops:
        for n in (N, N - 13, 1 << 20, 12345):
            try:
                blk = pkg.clMathOp(dt, 1, 2, 0, 0, getattr(pkg, op))
                av = a.view(-1)[:n * (w // 4)]; bv = b.view(-1)[:n * (w // 4)]; cv = c.view(-1)[:n * (w // 4)]
                if dt == pkg.DTYPE_INT: av = av.view(torch.int32); bv = bv.view(torch.int32); cv = cv.view(torch.int32)
                d = ev(lambda: blk.work_device(n, [av, bv], [cv]))
                print("clMathOp %-8s %-22s n=%9d: %7.1f us  %5.2f TB/s" % (name, op, n, d * 1e6, 3 * n * w / d / 1e12), flush=True)
            except Exception as ex:
                print("clMathOp %s %s n=%d: %s" % (name, op, n, str(ex)[:90]), flush=True)
