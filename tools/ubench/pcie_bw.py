"""Raw PCIe rates with pinned memory (the ceiling of the host path): H2D alone, D2H alone, both at once on two streams."""
import time
import torch
n = 256 << 20
h_in = torch.empty(n, dtype=torch.uint8).pin_memory()
h_out = torch.empty(n, dtype=torch.uint8).pin_memory()
d_a = torch.empty(n, dtype=torch.uint8, device="cuda")
d_b = torch.empty(n, dtype=torch.uint8, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def run(fn, it=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(it):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / it


def h2d():
    with torch.cuda.stream(s1):
        d_a.copy_(h_in, non_blocking=True)


def d2h():
    with torch.cuda.stream(s2):
        h_out.copy_(d_b, non_blocking=True)


def both():
    h2d(); d2h()


for chunk in (n,):
    print("H2D alone %.1f GB/s" % (n / run(h2d) / 1e9))
    print("D2H alone %.1f GB/s" % (n / run(d2h) / 1e9))
    print("both at once: %.1f GB/s each way" % (n / run(both) / 1e9))
# 8 MiB pieces (the host path's chunk size)
c = 8 << 20


def chunks():
    for i in range(0, n, c):
        with torch.cuda.stream(s1):
            d_a[i:i + c].copy_(h_in[i:i + c], non_blocking=True)
        with torch.cuda.stream(s2):
            h_out[i:i + c].copy_(d_b[i:i + c], non_blocking=True)


print("both at once in 8 MiB pieces: %.1f GB/s each way" % (n / run(chunks) / 1e9))
