"""clXEngine launches that want every CU next to another stream's kernel that holds K of them (the sharded pipeline's exchange runs on a side stream
under the correlation; a workgroup of the fused kernel needs a whole CU's registers, so it cannot share one).  Compares the in-launch reduction
with the second-kernel form and checks the results.  Needs tools/ubench/spin_kernel.co (hipcc --genco).  Tuning aid."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as e
pkg = e.load_package(); o = e.load_oracle()
hip = C.CDLL("libamdhip64.so")
mod, fn = C.c_void_p(), C.c_void_p()
co = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ubench", "spin_kernel.co")
assert hip.hipModuleLoad(C.byref(mod), co.encode()) == 0
assert hip.hipModuleGetFunction(C.byref(fn), mod, b"k_spin") == 0
side = torch.cuda.Stream()
sink = torch.zeros(16, dtype=torch.int32, device="cuda")


def hold(k, seconds):
    ticks, outp = C.c_ulonglong(int(seconds * 1e8)), C.c_void_p(sink.data_ptr())
    args = (C.c_void_p * 2)(C.cast(C.byref(ticks), C.c_void_p), C.cast(C.byref(outp), C.c_void_p))
    assert hip.hipModuleLaunchKernel(fn, k, 1, 1, 1024, 1, 1, 0, C.c_void_p(side.cuda_stream), args, None) == 0


def run(F, nint, k, rs):
    os.environ["MI355_XE_INKERNEL_REDUCE"] = "1" if rs else "0"
    N, T = 64, 1024
    xe = pkg.clXEngine(1, 2, 0, 0, False, pkg.DTYPE_BYTE, 1, N, 1, 0, F, T, [])
    x = torch.randint(-128, 128, (nint, T, N, F, 1, 2), dtype=torch.int8, device="cuda")
    out = torch.zeros(nint * xe.get_output_buffer_size(), 2, device="cuda")
    f = (lambda: xe.xcorrelate_n_device(nint, x, out)) if nint > 1 else (lambda: xe.xcorrelate_device(x, out))
    for _ in range(3): f()
    torch.cuda.synchronize()
    if k: hold(k, 0.05)                       # 50 ms on the side stream; the launches below run under it
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(100): f()
    b.record(); b.synchronize()
    dt = a.elapsed_time(b) * 10
    torch.cuda.synchronize()
    per = xe.get_output_buffer_size()
    ch = min(F, 16)  # (the oracle on the first 16 channels of the first window: channels are independent)
    ref = o.xengine_ichar(N, ch, 1, T, np.ascontiguousarray(x[0, :, :, :ch].cpu().numpy()).reshape(-1), exact=True)
    ok = np.array_equal(out[:per].cpu().numpy().view(np.complex64).reshape(F, -1)[:ch].reshape(-1), ref)
    print("F=%4d windows=%d, %2d CUs held, %-16s: %6.1f us per launch  %s" % (F, nint, k, "in-launch" if rs else "second kernel", dt, "bit exact" if ok else "MISMATCH"))


for F, nint in ((1024, 1), (128, 8)):
    for k in (0, 8, 32):
        for rs in (False, True):
            run(F, nint, k, rs)
