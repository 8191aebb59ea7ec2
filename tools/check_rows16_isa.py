#!/usr/bin/env python3
"""Static check of the instantiations of k_conv3x3_c64_rows16_bf16 that launch_igemm dispatches (KIND 0 / 1, with and without a residual).

Its MFMAs are inline asm (the weight fragments are pinned to AGPRs / VGPRs by constraint), so hipcc's hazard recogniser does not see them.
That is safe only while the register allocator keeps every weight fragment where it was pinned: a fragment it parks elsewhere is copied into
an AGPR quad in front of the MFMA that takes it (`v_accvgpr_write` / `v_accvgpr_mov` inside the row loop) WITHOUT the wait states between the
copy and the MFMA's operand read -- wrong, run-to-run different sums (seen in round 5: `profiles/r05_rows16_agpr_copy_hazard.txt`).  This tool
compiles sd_conv_rows16.hip to ISA (device side only, a few seconds) and fails when, behind the first MFMA of a dispatched instantiation,
  * an AGPR that any MFMA reads as its weight operand is written, or
  * a scratch (spill) instruction appears.
usage: check_rows16_isa.py [--keep file.s] [--all]"""
import re
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
DISPATCHED = ["ILi0ELb0E", "ILi0ELb1E", "ILi1ELb0E", "ILi1ELb1E"]


def check(asm_text):
    problems = []
    for inst in DISPATCHED:
        name = f"_ZN2sd25k_conv3x3_c64_rows16_bf16{inst}EEvNS_8RowsArgsE"
        m = re.search(r"^" + re.escape(name) + r":.*?s_endpgm", asm_text, re.S | re.M)
        if not m:
            problems.append(f"{inst}: instantiation not found in the ISA")
            continue
        lines = m.group(0).split("\n")
        mf = [i for i, l in enumerate(lines) if "v_mfma_f32_16x16x32_bf16" in l]
        if len(mf) != 432:
            problems.append(f"{inst}: {len(mf)} MFMAs, expected 432 (three row bodies of 144)")
            continue
        body = lines[mf[0]:]
        weight_regs, written = set(), set()
        for l in body:
            mm = re.search(r"v_mfma_f32_16x16x32_bf16 a\[\d+:\d+\], a\[(\d+):(\d+)\]", l)
            if mm:
                weight_regs.update(range(int(mm.group(1)), int(mm.group(2)) + 1))
            mm = re.search(r"v_accvgpr_(?:write|mov)_b32 a(\d+),", l)
            if mm:
                written.add(int(mm.group(1)))
        clash = sorted(weight_regs & written)
        if clash:
            problems.append(f"{inst}: AGPRs {clash[:8]}... are written inside the row loop AND read as MFMA weight operands")
        n_scratch = sum("scratch_" in l for l in body)
        if n_scratch:
            problems.append(f"{inst}: {n_scratch} scratch (spill) instructions behind the first MFMA")
    return problems


def check_old_kernels():
    """`--all`: the 32x32x16 bf16 row-stream kernel of sd_conv.hip pins 56 weight fragments by "a" constraints WITHOUT the tied re-definition; it has
    been right since round 2 because hipcc happens to keep them in place -- the same static check, on the big translation unit (~35 s)."""
    problems = []
    with tempfile.TemporaryDirectory() as tmp:
        out = Path(tmp) / "sd_conv.s"
        subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-x", "hip", "--cuda-device-only", "-S",
                        "-I", str(ROOT / "include"), str(ROOT / "structuredetector_amd" / "csrc" / "sd_conv.hip"), "-o", str(out)],
                       check=True, cwd=tmp, capture_output=True)
        text = out.read_text()
    m = re.search(r"^_ZN2sd23k_conv3x3_c64_rows_bf16ENS_8RowsArgsE:.*?s_endpgm", text, re.S | re.M)
    if not m:
        return ["k_conv3x3_c64_rows_bf16 not found in the ISA of sd_conv.hip"]
    lines = m.group(0).split("\n")
    mf = [i for i, l in enumerate(lines) if "v_mfma_f32_32x32x16_bf16" in l]
    body = lines[mf[0]:]
    weight, written = set(), set()
    for l in body:
        mm = re.search(r"v_mfma_f32_32x32x16_bf16 a\[\d+:\d+\], v\[\d+:\d+\], a\[(\d+):(\d+)\]", l)
        if mm:
            weight.update(range(int(mm.group(1)), int(mm.group(2)) + 1))
        mm = re.search(r"v_accvgpr_(?:write|mov)_b32 a(\d+),", l)
        if mm:
            written.add(int(mm.group(1)))
    if not weight:
        problems.append("k_conv3x3_c64_rows_bf16: no AGPR weight operands found (pattern out of date?)")
    if weight & written:
        problems.append(f"k_conv3x3_c64_rows_bf16: AGPRs {sorted(weight & written)[:8]}... written inside the row loop AND read as MFMA weight operands")
    print(f"k_conv3x3_c64_rows_bf16: {len(mf)} MFMAs, {len(weight)} weight AGPRs, {len(weight & written)} of them written inside the loop")
    return problems


def main():
    keep = sys.argv[sys.argv.index("--keep") + 1] if "--keep" in sys.argv else None
    with tempfile.TemporaryDirectory() as tmp:
        out = Path(keep) if keep else Path(tmp) / "sd_conv.s"
        subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-x", "hip", "--cuda-device-only", "-S",
                        "-I", str(ROOT / "include"), str(ROOT / "structuredetector_amd" / "csrc" / "sd_conv_rows16.hip"), "-o", str(out)],
                       check=True, cwd=tmp, capture_output=True)
        problems = check(out.read_text())
    if "--all" in sys.argv:
        problems += check_old_kernels()
    for pb in problems:
        print("FAIL", pb)
    print("k_conv3x3_c64_rows16_bf16:", "ok -- no weight AGPR is written and nothing is spilled inside the row loops" if not problems else f"{len(problems)} problem(s)")
    return 1 if problems else 0


if __name__ == "__main__":
    raise SystemExit(main())
