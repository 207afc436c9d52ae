"""gpurun_out/<tag>_cfg<k>_* (tools/make_profiles_cfg.sh) -> profiles/<tag>_baseline_configs.txt (per BASELINE config: the kernel's average
duration in the trace of a command that launches ONLY that size, FETCH_SIZE / WRITE_SIZE per dispatch, the ratio to the algorithmic bytes) and
profiles/baseline_configs_pmc.json (the traffic ratios bench.py quotes under `baseline_configs`).
usage: python tools/collect_profiles_cfg.py <tag>"""
import json
import os
import sys

tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(root, "gpurun_out", ""), os.path.join(root, "profiles", "")
KERNEL = {"2": "k_fft<4096", "3": "k_ols<", "4": "k_pfbw<", "5": "k_xe_i8_", "5b": "k_xe_i8_"}


def rows(path, prefix):
    out = []
    try:
        for l in open(path):
            if l.startswith(prefix):
                out.append(l.rstrip("\n"))
    except OSError:
        pass
    return out


pmc = {}
with open(P + tag + "_baseline_configs.txt", "w") as o:
    o.write("# One BASELINE config per command (tools/baseline_cfg.py <k>: only that config's benchmark-size launch), three separate passes each:\n"
            "#   rocprofv3 --kernel-trace --stats | rocprofv3 --pmc FETCH_SIZE | rocprofv3 --pmc WRITE_SIZE      (tools/make_profiles_cfg.sh %s)\n"
            "# FETCH_SIZE / WRITE_SIZE are KiB per dispatch; on gfx950 FETCH_SIZE counts a 128-byte request as 64 B, so read bytes = 2 x FETCH_SIZE x 1024\n"
            "# (MI355X_MICROARCH.md, HBM section).  frac = algorithmic bytes / trace average / 8 TB/s; traffic ratio = (2 x FETCH + WRITE) / algorithmic.\n" % tag)
    for k in ("2", "3", "4", "5", "5b"):
        try:
            run = json.loads(open(G + "%s_cfg%s_run.json" % (tag, k)).read().strip().splitlines()[-1])
        except (OSError, IndexError, ValueError):
            continue
        st = rows(G + "%s_cfg%s_stats.txt" % (tag, k), KERNEL[k])
        fe = [l for l in rows(G + "%s_cfg%s_fetch.txt" % (tag, k), KERNEL[k]) if "FETCH_SIZE" in l]
        wr = [l for l in rows(G + "%s_cfg%s_write.txt" % (tag, k), KERNEL[k]) if "WRITE_SIZE" in l]
        o.write("\n== config %s: %s\n" % (k, run["what"]))
        o.write("unprofiled run, HIP events over %d launches: %.2f us per launch -> %.4f of 8 TB/s on %d algorithmic bytes\n"
                % (run["launches"], run["us_per_launch"], run["hbm_frac"], run["algorithmic_bytes_per_launch"]))
        o.write("%-92s %6s %12s %12s %7s\n" % ("kernel (trace)", "calls", "total_us", "avg_us", "pct"))
        for l in st:
            o.write(l + "\n")
        o.write("%-92s %-14s %6s %16s %12s\n" % ("kernel (pmc)", "counter", "n", "avg_value", "avg_dur_us"))
        for l in fe + wr:
            o.write(l + "\n")
        if st and fe and wr:
            # the largest launch of the family is the benchmark launch (block construction may launch small tuning kernels of the same name)
            big = lambda ls, col: max(float(l.split()[col]) for l in ls)
            avg_us = float(max(st, key=lambda l: float(l.split()[-3])).split()[-2])
            f, w = big(fe, -2), big(wr, -2)
            moved = (2 * f + w) * 1024
            alg = run["algorithmic_bytes_per_launch"]
            o.write("-> trace average %.2f us = %.4f of 8 TB/s; moved %.0f B = %.3f x algorithmic (read %.0f B, written %.0f B)\n"
                    % (avg_us, alg / (avg_us * 1e-6) / 8e12, moved, moved / alg, 2 * f * 1024, w * 1024))
            pmc[k] = {"trace_avg_us": avg_us, "traffic_ratio": round(moved / alg, 4), "hbm_bytes_per_launch": int(moved),
                      "algorithmic_bytes_per_launch": alg}
pmc["source"] = "profiles/%s_baseline_configs.txt" % tag
json.dump(pmc, open(P + "baseline_configs_pmc.json", "w"), indent=1)
print(json.dumps(pmc))
