"""Turn the scratch output of tools/make_profiles.sh <tag> (gpurun_out/<tag>_*) into the committed evidence under profiles/:
<tag>_bench.json, <tag>_kernel_stats.txt, <tag>_traffic.txt and fft4096_pmc.json (the traffic figure bench.py reports).
usage: python tools/collect_profiles.py <tag>"""
import json, os, sys

tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(root, "gpurun_out", ""), os.path.join(root, "profiles", "")


def counters(path):
    out, on, hdr = [], False, ""
    for l in open(path):
        if l.startswith("kernel") and "counter" in l:
            on, hdr = True, l.rstrip("\n")
            continue
        if on and l.strip():
            out.append(l.rstrip("\n"))
    return hdr, out


h, f = counters(G + tag + "_fetch.txt")
_, w = counters(G + tag + "_write.txt")
with open(P + tag + "_traffic.txt", "w") as o:
    o.write("# HBM traffic per launch from PMC counters, separate passes (rocprofv3 --pmc FETCH_SIZE ; rocprofv3 --pmc WRITE_SIZE), "
            "bench.py --steps 5 --warmup 1 --no-cpu --no-sustained  (tools/make_profiles.sh %s)\n" % tag)
    o.write("# values are KiB per dispatch; on gfx950 FETCH_SIZE counts 128-B read requests as 64 B, so HBM read bytes = 2 x FETCH_SIZE x 1024 "
            "(MI355X_MICROARCH.md, HBM section)\n")
    o.write(h + "\n" + "\n".join(f) + "\n" + "\n".join(w) + "\n")
with open(P + tag + "_kernel_stats.txt", "w") as o:
    o.write("# command: rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 3 --no-cpu --sustain-s 0.5   (tools/make_profiles.sh %s)\n" % tag)
    o.write("# bench.py's own line from the same box, unprofiled run: profiles/%s_bench.json (roofline.kernel_us = HIP-event time over the K launches / K)\n" % tag)
    o.write(open(G + tag + "_stats.txt").read())
open(P + tag + "_bench.json", "w").write(open(G + tag + "_bench.json").read().strip().splitlines()[-1] + "\n")


def val(lines):
    # the headline launches (16384 frames) and the one-vector host-path calls run the same kernel family: take the large one
    return max(float(l.split()[-2]) for l in lines if l.startswith("k_fft<4096"))


fs, ws = val(f), val(w)
d = {"kernel": "k_fft<4096,-1,false,1,Geo<4096>>",
     "command": "bench.py --steps 5 --warmup 1 --no-cpu (tools/make_profiles.sh, separate --pmc FETCH_SIZE and --pmc WRITE_SIZE passes)",
     "FETCH_SIZE_KiB_per_launch": fs, "WRITE_SIZE_KiB_per_launch": ws, "fetch_correction": 2.0,
     "hbm_bytes_per_launch": int(round((2 * fs + ws) * 1024)), "algorithmic_bytes_per_launch": 1073741824,
     "source": "profiles/%s_traffic.txt" % tag}
json.dump(d, open(P + "fft4096_pmc.json", "w"))
b = json.loads(open(P + tag + "_bench.json").read())
print("value %.1f %s  roofline.frac %.4f  kernel_us %.2f  traffic %d B" % (b["value"], b["unit"], b["roofline"]["frac"], b["roofline"]["kernel_us"], d["hbm_bytes_per_launch"]))
