"""Assembles profiles/<tag>_xengine.txt from the gpurun_out/<tag>_xe_* files written by tools/make_profiles_xe.sh."""
import sys

tag = sys.argv[1]
O = "gpurun_out/"


def val(fn, pat, ctr):
    for ln in open(fn):
        if pat in ln and ctr in ln:
            return float(ln.split()[-2])
    raise SystemExit("%s: no %s line for %s" % (fn, ctr, pat))


def val_or(fn, pat, ctr):
    try:
        return val(fn, pat, ctr)
    except SystemExit:
        return None


# config 5 is ONE kernel since round 4 (the time ranges are combined by the fused kernel's own reduce-scatter tail); older tags: fused + reduce
K0 = "k_xe_i8_lines<true, 1>"  # round 6: one window per call = the whole-line kernel with four time ranges combined inside the launch
K1 = "k_xe_i8_fused<1, 4, true, true, true>"
f0, w0 = val_or(O + tag + "_xe_fetch.txt", K0, "FETCH_SIZE"), val_or(O + tag + "_xe_write.txt", K0, "WRITE_SIZE")
f1, w1 = val_or(O + tag + "_xe_fetch.txt", K1, "FETCH_SIZE"), val_or(O + tag + "_xe_write.txt", K1, "WRITE_SIZE")
if f0 is not None:
    tot = (2 * f0 + w0) * 1024 / 1e6
    traffic = "# config 5, one window per call: 2 x %.2f + %.2f KiB (ONE kernel, k_xe_i8_lines<true, 1>: whole-line requests, four time ranges per team, their partial sums exchanged and combined inside the launch) = %.1f MB = %.2f x the 151.3 MB algorithmic bytes at the L2's memory side (98 MB of it the exchange; round 5's 32-byte-slice kernel: 1.63 x).  NOTE: this command set launches the same kernel for other geometries too, the average mixes them -- config 5 alone, one launch size per command: profiles/%s_baseline_configs.txt (1.80 x)" % (f0, w0, tot, tot / 151.3, tag)
elif f1 is not None:
    tot = (2 * f1 + w1) * 1024 / 1e6
    traffic = "# config 5: 2 x %.2f + %.2f KiB (ONE kernel: corner turn, correlation and the reduce-scatter of the four time ranges) = %.1f MB = %.2f x the 151.3 MB algorithmic bytes at the L2's memory side (the early touches of the slow lines count twice here, HBM reads them once: 1.50 x without them; round 3: 1.70 x, round 1: 2.78 x)" % (f1, w1, tot, tot / 151.3)
else:
    f1, w1 = val(O + tag + "_xe_fetch.txt", "k_xe_i8_fused<1, 4, true>", "FETCH_SIZE"), val(O + tag + "_xe_write.txt", "k_xe_i8_fused<1, 4, true>", "WRITE_SIZE")
    f2, w2 = val(O + tag + "_xe_fetch.txt", "k_xe_i8_reduce<1, 4>", "FETCH_SIZE"), val(O + tag + "_xe_write.txt", "k_xe_i8_reduce<1, 4>", "WRITE_SIZE")
    tot = (2 * f1 + w1 + 2 * f2 + w2) * 1024 / 1e6
    traffic = "# config 5: 2 x %.2f + %.2f (fused) + 2 x %.2f + %.2f (reduce) KiB = %.1f MB = %.2f x the 151.3 MB algorithmic bytes (round 1: 2.78 x)" % (f1, w1, f2, w2, tot, tot / 151.3)
out = ["# X-engine IChar path alone, BASELINE config 5 (64 ant x 1024 ch x 1024 frames) and its dual-polarisation sibling (32 ant), device resident,",
       "# back-to-back launches over >= 640 MB of distinct inputs in rotation (every launch reads its input from HBM; rounds 1-3 and the first r04 files re-read",
       "# ONE buffer, which the 256 MiB Infinity Cache serves: 55 us at config 5 against 86 from HBM then, 64 from HBM since the early touches of the slow lines): tools/make_profiles_xe.sh %s  (two command sets, each with its own passes; 20 launches right after idle, so the times" % tag,
       "# here are a few per cent above the bench's).  SQ_VALU_MFMA_BUSY_CYCLES = 16 cycles per v_mfma_i32_16x16x64_i8 / _16x16x32_i8, summed over the 1024 SIMDs.",
       "# pass 1: rocprofv3 --kernel-trace --stats ; passes 2-4: --pmc FETCH_SIZE / --pmc WRITE_SIZE / --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE (separate runs)",
       "# FETCH_SIZE / WRITE_SIZE are KiB per dispatch; gfx950: HBM read bytes = 2 x FETCH_SIZE x 1024 (MI355X_MICROARCH.md, HBM section)",
       traffic,
       "", "== probe output (HIP-event time per integration)"]
def section(setname, title):
    o = ["", "==== " + title, "== probe output (HIP-event time per call)"]
    o += [ln.rstrip() for ln in open(O + tag + "_" + setname + "_stats.log") if "clXEngine" in ln or "per-rank" in ln or "single GPU" in ln]
    o += ["", "== kernel trace (avg_us per dispatch)"]
    o += [ln.rstrip() for ln in open(O + tag + "_" + setname + "_stats.txt") if ln.startswith("kernel") or ln.startswith("k_xe")]
    o += ["", "== counters (avg per dispatch)"]
    for k in ("fetch", "write", "mfma"):
        o += [ln.rstrip() for ln in open(O + tag + "_" + setname + "_" + k + ".txt")
              if ln.startswith("k_xe") and any(c in ln for c in ("FETCH", "WRITE", "MFMA", "BUSY", "GUI"))]
    return o


out = out[:-2]
out += section("xe", "BASELINE config 5 (64 ant x 1024 ch x 1024 frames) and its siblings: python tools/probe.py rates xengine")
out += section("xl", "rows > 64 (k_xe_turn_lds + k_xe_corr_sb / k_xe_corr_lds) and the per-rank 128-channel problem, one window per launch and batched "
                     "(kernel names are shared by the geometries of this set: the per-dispatch averages mix them): python tools/xe_large.py")
open("profiles/" + tag + "_xengine.txt", "w").write("\n".join(out) + "\n")
print("profiles/%s_xengine.txt: %.1f MB = %.2f x algorithmic" % (tag, tot, tot / 151.3))
