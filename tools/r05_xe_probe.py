"""clXEngine config 5 from HBM (inputs in rotation): A/B of the tuning switches of xengine_fused.hip inside ONE process (round 5 tuning aid).
usage: python tools/r05_xe_probe.py "NAME=VAL,NAME=VAL" ...   each argument is one arm (comma-separated environment settings; "base" = none);
PROBE_NINT windows per launch (1), PROBE_NBUF inputs in rotation (4), PROBE_IT rounds (30), PROBE_CHECK=1 compares every arm's output with the first arm's."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as e
pkg = e.load_package()
N, F, T = int(os.environ.get("PROBE_N", "64")), int(os.environ.get("PROBE_F", "1024")), int(os.environ.get("PROBE_T", "1024"))
nint = int(os.environ.get("PROBE_NINT", "1")); nbuf = int(os.environ.get("PROBE_NBUF", "4")); it = int(os.environ.get("PROBE_IT", "30"))
check = int(os.environ.get("PROBE_CHECK", "0"))
arms = sys.argv[1:] or ["base"]
xe = pkg.clXEngine(1, 2, 0, 0, False, pkg.DTYPE_BYTE, 1, N, 1, 0, F, T, [])
win = T * N * F * 2
maxpad = max([int(kv.split("=")[1]) for a in arms for kv in a.split(",") if kv.startswith("MI355_XE_WINPAD=")] + [0])
g = torch.Generator(device="cuda"); g.manual_seed(5)
xs = [torch.randint(-127, 128, (nint * (win + maxpad),), dtype=torch.int8, device="cuda", generator=g) for _ in range(nbuf)]
outs = [torch.zeros(nint * xe.get_output_buffer_size(), 2, device="cuda") for _ in range(nbuf)]
def fill_padded(pad):  # window w of buffer k at offset w * (win + pad): the same samples for every pad
    if nint == 1 or maxpad == 0: return
    for k in range(nbuf):
        gk = torch.Generator(device="cuda"); gk.manual_seed(100 + k)
        for w in range(nint):
            xs[k][w * (win + pad): w * (win + pad) + win] = torch.randint(-127, 128, (win,), dtype=torch.int8, device="cuda", generator=gk)
def run():
    for k in range(nbuf):
        if nint > 1: xe.xcorrelate_n_device(nint, xs[k], outs[k])
        else: xe.xcorrelate_device(xs[k], outs[k])
ref = None
for arm in arms:
    sets = [] if arm == "base" else [kv.split("=") for kv in arm.split(",")]
    for k, v in sets: os.environ[k] = v
    pad = int(dict(sets).get("MI355_XE_WINPAD", 0))
    fill_padded(pad)
    for _ in range(3): run()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): run()
    b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) * 1e3 / (it * nbuf)
    msg = ""
    if check:
        cur = [o.clone() for o in outs]
        if ref is None: ref = cur
        else: msg = "  outputs equal to the first arm's: %s" % all(torch.equal(x, y) for x, y in zip(ref, cur))
    print("%-60s %8.1f us per launch  %7.2f us per window%s" % (arm, us, us / nint, msg), flush=True)
    for k, v in sets: del os.environ[k]
