"""Summarise -Rpass-analysis=kernel-resource-usage for one .hip file: name, VGPR, spill, LDS, occupancy."""
import re, subprocess, sys
src = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else ""
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fno-slp-vectorize", "-Iinclude", "-Igr-clenabled_amd/csrc", "-c", src,
       "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = {}
rows = []
for line in out.splitlines():
    m = re.search(r"remark:\s+(Function Name|VGPRs|AGPRs|VGPRs Spill|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]|TotalSGPRs):\s*(\S+)", line)
    if not m: continue
    k, v = m.group(1), m.group(2)
    if k == "Function Name":
        cur = {"name": v}; rows.append(cur)
    else: cur[k] = v
for r in rows:
    n = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    n = re.sub(r"\(anonymous namespace\)::", "", n)[:70]
    if pat and pat not in n: continue
    print("%-70s vgpr=%-4s sgpr=%-4s spill=%-4s scratch=%-5s lds=%-6s occ=%s" % (n, r.get("VGPRs"), r.get("TotalSGPRs"), r.get("VGPRs Spill"),
          r.get("ScratchSize [bytes/lane]"), r.get("LDS Size [bytes/block]"), r.get("Occupancy [waves/SIMD]")))
