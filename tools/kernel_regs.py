#!/usr/bin/env python3
"""Register / scratch budget and a few instruction counts of the kernels in one built object.
usage: python tools/kernel_regs.py gr-clenabled_amd/csrc/build/fft_mr.o [name filter] [instruction ...]"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
obj = os.path.abspath(sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
ops = sys.argv[3:]
tmp = tempfile.mkdtemp()
try:
    shutil.copy(obj, os.path.join(tmp, "x.o"))
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", "x.o"], cwd=tmp, check=True, capture_output=True)
    co = [f for f in os.listdir(tmp) if "gfx950" in f][0]
    notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], cwd=tmp, check=True, capture_output=True, text=True).stdout
    for blk in notes.split("  - .agpr_count:")[1:]:
        f = {m.group(1): m.group(2) for m in re.finditer(r"\.(\w+):\s+(\S+)", ".agpr_count:" + blk)}
        if flt in f["name"]:
            print(f["name"][:110], "vgpr", f["vgpr_count"], "agpr", f["agpr_count"], "sgpr", f.get("sgpr_count"), "spill", f["vgpr_spill_count"], "scratch",
                  f["private_segment_fixed_size"], "lds", f.get("group_segment_fixed_size"))
    if ops:
        asm = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", co], cwd=tmp, check=True, capture_output=True, text=True).stdout
        cur, cnt = None, {}
        for line in asm.splitlines():
            m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
            if m:
                cur = m.group(1)
                continue
            if cur and flt in cur:
                parts = line.split()
                op = re.sub(r"_(e32|e64|dpp|sdwa)$", "", parts[0]) if parts else ""
                if op in ops or (ops == ["all"] and op and not op.endswith(":")):
                    cnt.setdefault(cur, {}).setdefault(op, 0)
                    cnt[cur][op] += 1
        for k, v in cnt.items():
            print(k[:110], v)
finally:
    shutil.rmtree(tmp, ignore_errors=True)
