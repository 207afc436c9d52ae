"""Assembles profiles/<tag>_xengine.txt from the gpurun_out/<tag>_xe_* files written by tools/make_profiles_xe.sh."""
import sys

tag = sys.argv[1]
O = "gpurun_out/"


def val(fn, pat, ctr):
    for ln in open(fn):
        if pat in ln and ctr in ln:
            return float(ln.split()[-2])
    raise SystemExit("%s: no %s line for %s" % (fn, ctr, pat))


f1, w1 = val(O + tag + "_xe_fetch.txt", "k_xe_i8_fused<1, 4, true>", "FETCH_SIZE"), val(O + tag + "_xe_write.txt", "k_xe_i8_fused<1, 4, true>", "WRITE_SIZE")
f2, w2 = val(O + tag + "_xe_fetch.txt", "k_xe_i8_reduce<1, 4>", "FETCH_SIZE"), val(O + tag + "_xe_write.txt", "k_xe_i8_reduce<1, 4>", "WRITE_SIZE")
tot = (2 * f1 + w1 + 2 * f2 + w2) * 1024 / 1e6
out = ["# X-engine IChar path alone, BASELINE config 5 (64 ant x 1024 ch x 1024 frames) and its dual-polarisation sibling (32 ant), device resident,",
       "# back-to-back launches: tools/make_profiles_xe.sh %s  (command under rocprofv3: python tools/probe.py rates xengine -- 20 launches right after idle," % tag,
       "# so the times here are a few per cent above the bench's)",
       "# pass 1: rocprofv3 --kernel-trace --stats ; passes 2-4: --pmc FETCH_SIZE / --pmc WRITE_SIZE / --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE (separate runs)",
       "# FETCH_SIZE / WRITE_SIZE are KiB per dispatch; gfx950: HBM read bytes = 2 x FETCH_SIZE x 1024 (MI355X_MICROARCH.md, HBM section)",
       "# config 5: 2 x %.2f + %.2f (fused) + 2 x %.2f + %.2f (reduce) KiB = %.1f MB = %.2f x the 151.3 MB algorithmic bytes (round 1: 2.78 x)" % (f1, w1, f2, w2, tot, tot / 151.3),
       "", "== probe output (HIP-event time per integration = fused kernel + reduce kernel)"]
out += [ln.rstrip() for ln in open(O + tag + "_xe_stats.log") if "clXEngine" in ln]
out += ["", "== kernel trace (avg_us per dispatch)"]
out += [ln.rstrip() for ln in open(O + tag + "_xe_stats.txt") if ln.startswith("kernel") or ln.startswith("k_xe")]
out += ["", "== counters (avg per dispatch)"]
for k in ("fetch", "write", "mfma"):
    out += [ln.rstrip() for ln in open(O + tag + "_xe_" + k + ".txt")
            if (ln.startswith("k_xe_i8") or ln.startswith("k_xe_f32_fused")) and any(c in ln for c in ("FETCH", "WRITE", "MFMA", "BUSY", "GUI"))]
open("profiles/" + tag + "_xengine.txt", "w").write("\n".join(out) + "\n")
print("profiles/%s_xengine.txt: %.1f MB = %.2f x algorithmic" % (tag, tot, tot / 151.3))
