import os, sys
sys.path.insert(0, "/root/repo")
import torch, __graft_entry__ as e
pkg = e.load_package()
N, F, T = 64, 1024, 1024
xe = pkg.clXEngine(1, 2, 0, 0, False, pkg.DTYPE_BYTE, 2, N, 1, 0, F, T, [])
g = torch.Generator(device="cuda").manual_seed(1)
bufs = [torch.randint(-127, 128, (T, N, F, 2, 2), dtype=torch.int8, device="cuda", generator=g) for _ in range(4)]
vis = torch.zeros(xe.get_output_buffer_size(), 2, device="cuda")
for i in range(50): xe.xcorrelate_device(bufs[i % 4], vis)
torch.cuda.synchronize()
os.environ["MI355_XE_TS"] = "1"
for i in range(4): xe.xcorrelate_device(bufs[i % 4], vis)
