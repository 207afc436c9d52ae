"""Build gate for csrc/xengine_lines.hip (no GPU needed): k_xe_i8_lines keeps its 128 accumulators in a0..a127 BY NAME across separate inline-assembly
statements; the compiler only sees them as clobbers, which forbids values living across an asm from sitting there but not short-lived values between two
asms.  The guard is the per-object flag -amdgpu-spill-vgpr-to-agpr=0.  This test disassembles the built object and fails if anything but the
hand-written v_mfma / v_accvgpr instructions touches an accumulation register, if the flag can be dropped by `make CXXFLAGS=...`, or if the register
/ scratch budget of the kernels moves.  (The bit-exact comparison against k_xe_i8_fused in tests/test_xengine_lines_gpu.py is the other half of the gate.)"""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "gr-clenabled_amd", "csrc")
OBJ = os.path.join(CSRC, "build", "xengine_lines.o")
LLVM = "/opt/rocm/lib/llvm/bin"


@pytest.fixture(scope="module")
def device_object():
    if not os.path.exists(OBJ):
        import sys
        sys.path.insert(0, ROOT)
        import __graft_entry__ as e
        e.build()
    tmp = tempfile.mkdtemp()
    try:
        shutil.copy(OBJ, os.path.join(tmp, "x.o"))
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", "x.o"], cwd=tmp, check=True, capture_output=True)
        co = [f for f in os.listdir(tmp) if "gfx950" in f]
        assert len(co) == 1, os.listdir(tmp)
        yield os.path.join(tmp, co[0])
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def test_only_hand_written_instructions_touch_the_accumulators(device_object):
    asm = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", device_object], check=True, capture_output=True, text=True).stdout
    allowed = {"v_mfma_i32_16x16x64_i8", "v_accvgpr_read_b32", "v_accvgpr_write_b32"}
    seen, bad = {}, []
    for line in asm.splitlines():
        parts = line.split("//")[0].split()
        if not parts or parts[0].endswith(":"):
            continue
        op, operands = parts[0], " ".join(parts[1:])
        if re.search(r"(?<![A-Za-z0-9_])a(\d+|\[\d+:\d+\])(?![A-Za-z0-9_])", operands):
            seen[op] = seen.get(op, 0) + 1
            if op not in allowed:
                bad.append(line.strip())
    assert not bad, "instructions the compiler generated on accumulation registers:\n" + "\n".join(bad[:20])
    # both kernels (plain and time-range form) x both bodies (diagonal / off-diagonal groups): 64 products' worth of accumulator references each
    assert seen.get("v_mfma_i32_16x16x64_i8", 0) >= 200 and seen.get("v_accvgpr_read_b32", 0) >= 4 * 128 and seen.get("v_accvgpr_write_b32", 0) >= 4 * 128, seen


def test_register_and_scratch_budget(device_object):
    notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", device_object], check=True, capture_output=True, text=True).stdout
    kernels = {}
    for blk in notes.split("  - .agpr_count:")[1:]:
        f = {m.group(1): m.group(2) for m in re.finditer(r"\.(\w+):\s+(\S+)", ".agpr_count:" + blk)}
        kernels[f["name"]] = f
    lines = {k: v for k, v in kernels.items() if "k_xe_i8_lines" in k}
    assert len(lines) == 3, list(kernels)  # plain, time ranges, two polarisations
    for name, f in lines.items():
        # all 128 accumulation registers + at most 128 vector registers (two waves per SIMD); a handful of spilled registers outside the K loop is
        # what the build has had since round 5 (12 / 48 bytes of scratch per lane) -- a jump means the allocator has started spilling in the loop
        # (vgpr_count = vector + accumulation registers of a wave: at most 256 for two waves per SIMD; the two-polarisation kernel needs 244)
        assert int(f["agpr_count"]) == 128 and 128 < int(f["vgpr_count"]) <= 256, (name, f)
        assert int(f["private_segment_fixed_size"]) <= 64 and int(f["vgpr_spill_count"]) <= 24, (name, f)


def test_the_flag_cannot_be_dropped_from_the_command_line():
    mk = open(os.path.join(CSRC, "Makefile")).read()
    assert re.search(r"^build/xengine_lines\.o:\s*override CXXFLAGS \+= .*-amdgpu-spill-vgpr-to-agpr=0", mk, re.M), "target-specific flag must use `override`"
    out = subprocess.run(["make", "-n", "-B", "-C", CSRC, "CXXFLAGS=-O1", "build/xengine_lines.o"], capture_output=True, text=True).stdout
    assert "-amdgpu-spill-vgpr-to-agpr=0" in out, out[-500:]


def test_only_the_lds_dma_statements_use_m0(device_object):
    """ln_dma16 writes m0 without restoring it (two scalar moves less per request in the hot loop).  m0 is reserved to the compiler, which does not track
    such a write: the object must therefore contain no instruction that READS m0 other than the save of the one asm statement that still restores it
    (the progress-word poll), i.e. every mention of m0 is `s_mov_b32 m0, <sgpr>` or `s_mov_b32 <sgpr>, m0`."""
    asm = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", device_object], check=True, capture_output=True, text=True).stdout
    bad = []
    for line in asm.splitlines():
        code = line.split("//")[0]
        if re.search(r"(?<![A-Za-z0-9_])m0(?![A-Za-z0-9_])", code):
            parts = code.replace(",", " ").split()
            ok = len(parts) == 3 and parts[0] == "s_mov_b32" and (parts[1] == "m0" or parts[2] == "m0")
            if not ok:
                bad.append(line.strip())
    assert not bad, "\n".join(bad[:20])
