"""Random geometries, window counts and switches through the whole-line X-engine kernel (csrc/xengine_lines.hip) against the 32-byte-slice kernel:
every window must be bit identical (both are bit exact against the oracle in tests/test_xengine_lines_gpu.py; this sweep looks for a geometry,
a share of units per workgroup, a line rotation, a touch distance or a pace that the tests do not hold).
usage: python tools/stress/xe_lines_fuzz.py [seconds] [seed]      (on the GPU box)"""
import os, sys, time, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import __graft_entry__ as e
pkg = e.load_package()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
N, t0, cases, lines_cases, bad = 64, time.time(), 0, 0, 0
SW = ("MI355_XE_LINES_PACE", "MI355_XE_LINES_ROT", "MI355_XE_LINES_PF", "MI355_XE_LINES_PUB", "MI355_XE_LINES_MAX_ITEMS", "MI355_XE_LINES_MIN_UNITS",
      "MI355_XE_LINES_SPLIT_ANY", "MI355_XE_TSPLIT", "MI355_XE_DBG", "MI355_XE_WAIT_US")
routes = {}
while time.time() - t0 < budget:
    F = 64 * rnd.choice([1, 2, 3, 4, 8, 8, 16, 16, 16, 32])
    T = 32 * rnd.choice([1, 2, 3, 5, 6, 8, 9, 16])
    nint = rnd.choice([1, 2, 3, 4, 5, 8, 8, 12, 16, 24, 32, 48])
    if nint * T * N * F * 2 > 1 << 30: continue
    env = {"MI355_XE_LINES_PACE": str(rnd.choice([0, 1, 2, 3, 5])), "MI355_XE_LINES_ROT": str(rnd.choice([0, 1, 1, 3, 7])),
           "MI355_XE_LINES_PF": str(rnd.choice([0, 1, 2, 4, 4, 7])), "MI355_XE_LINES_PUB": str(rnd.choice([0, 1, 1, 1])),
           "MI355_XE_LINES_MAX_ITEMS": str(rnd.choice([2, 4, 16, 16, 64]))}
    if rnd.random() < 0.5: env["MI355_XE_LINES_MIN_UNITS"] = str(rnd.choice([4, 32, 64]))
    # round 6: the time-range form (two or four ranges per team, combined inside the launch); now and then with every bounded wait run out at once
    # (the last arriver of a team finishes what the others handed over) or with a short wait
    if rnd.random() < 0.45:
        S = rnd.choice([2, 4])
        if T % (32 * S) == 0:
            env.pop("MI355_XE_LINES_MIN_UNITS", None)
            env["MI355_XE_LINES_SPLIT_ANY"] = "1"
            env["MI355_XE_TSPLIT"] = str(S)
            r = rnd.random()
            if r < 0.15: env["MI355_XE_DBG"] = "512"
            elif r < 0.3: env["MI355_XE_WAIT_US"] = str(rnd.choice([0, 1, 3]))
    for k in SW: os.environ.pop(k, None)
    os.environ.update(env)
    if rnd.random() < 0.2:
        # round 6: 64 stations x two polarisations (lines of 32 channels, eight pair groups per line) against corner turn + correlator
        F2 = 32 * rnd.choice([1, 2, 3, 4, 8, 16, 32])
        if T * N * F2 * 4 > 1 << 29: continue
        for k in SW: os.environ.pop(k, None)
        os.environ["MI355_XE_LINES_PF"] = env["MI355_XE_LINES_PF"]
        os.environ["MI355_XE_LINES_MAX_ITEMS"] = env["MI355_XE_LINES_MAX_ITEMS"]
        if rnd.random() < 0.8: os.environ["MI355_XE_LINES_MIN_UNITS"] = "8"
        xe = pkg.clXEngine(1, 2, 0, 0, False, pkg.DTYPE_BYTE, 2, N, 1, 0, F2, T, [])
        g = torch.Generator(device="cuda").manual_seed(cases)
        x = torch.randint(-128, 128, (T, N, F2, 2, 2), dtype=torch.int8, device="cuda", generator=g)
        a = torch.zeros(xe.get_output_buffer_size(), 2, device="cuda"); b = torch.zeros_like(a)
        xe.xcorrelate_device(x, a); torch.cuda.synchronize()
        rt = xe.last_route()["kernel"]
        routes[rt] = routes.get(rt, 0) + 1
        os.environ["MI355_XE_NO_LINES2"] = "1"
        xe.xcorrelate_device(x, b); torch.cuda.synchronize()
        os.environ.pop("MI355_XE_NO_LINES2")
        cases += 1
        if not torch.equal(a, b):
            bad += 1
            print("MISMATCH two polarisations F=%d T=%d %s" % (F2, T, dict(os.environ)), flush=True)
        del xe, x, a, b
        continue
    xe = pkg.clXEngine(1, 2, 0, 0, False, pkg.DTYPE_BYTE, 1, N, 1, 0, F, T, [])
    g = torch.Generator(device="cuda").manual_seed(cases)
    x = torch.randint(-128, 128, (nint, T, N, F, 1, 2), dtype=torch.int8, device="cuda", generator=g)
    per = xe.get_output_buffer_size()
    a = torch.zeros(nint * per, 2, device="cuda"); b = torch.zeros_like(a)
    xe.xcorrelate_n_device(nint, x, a)
    if rnd.random() < 0.5: xe.xcorrelate_n_device(nint, x, a)  # (a second launch on the same handle: tags, banks)
    torch.cuda.synchronize()
    rt = xe.last_route()["kernel"]
    routes[rt] = routes.get(rt, 0) + 1
    for k in ("MI355_XE_DBG", "MI355_XE_WAIT_US"): os.environ.pop(k, None)
    os.environ["MI355_XE_NO_LINES"] = "1"
    xe.xcorrelate_n_device(nint, x, b); torch.cuda.synchronize()
    os.environ.pop("MI355_XE_NO_LINES")
    cases += 1
    if not torch.equal(a, b):
        bad += 1
        print("MISMATCH F=%d T=%d nint=%d %s" % (F, T, nint, env), flush=True)
    del xe, x, a, b
print("xe_lines_fuzz: %d cases in %.0f s, %d mismatches; routes of the first launch form: %s" % (cases, time.time() - t0, bad, routes))
sys.exit(1 if bad else 0)
