#!/usr/bin/env python3
"""ISA-level check of the inline-asm operand allocation of csrc/blend_bwd.hip's pair_row_reduce9 / pair_row_reduce12.

The asm blocks take a[] as tied in/out operands ("+v") and b[] as plain inputs ("v"); b[] is READ (step 1, second half: the
``row_ror:8 ... bank_mask:0xc`` adds) AFTER every a[] has been WRITTEN (first half: ``bank_mask:0x3``).  Nothing in the constraints
forbids the register allocator from giving a b[k] the register of an a[j] -- it can only do so when both hold the same value, which
the callers never pass, but that is an argument, not a guarantee, and a compiler bump could break it silently (the formally safe
"+&v" costs 6-8 v_mov per entry pair: tools/probe/README.md).  So the guarantee is checked where it matters: in the code the
compiler actually produced.  This script compiles blend_bwd.hip to gfx950 assembly with build.sh's flags (no GPU needed), finds every
instance of the scheme (the two big reductions and the smaller joint reductions built the same way), and asserts that no source register of the second half is a destination register of the first half.

usage: python tools/check_reduce_regs.py        (exit code 0 = every instance is safe; prints the instances found)
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "4d-gaussian-splatting_amd", "csrc", "blend_bwd.hip")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
DPP = re.compile(r"^\s*v_add_f32_dpp\s+(v\d+),\s*(v\d+),\s*(v\d+)\s+row_ror:8\s+row_mask:0xf\s+bank_mask:(0x[0-9a-f]+)")


def assembly():
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "--cuda-device-only", "-S", "-o", "-", SRC,
           "-I", os.path.join(ROOT, "include")]
    return subprocess.run(cmd, check=True, capture_output=True, text=True).stdout


def instances(asm):
    """[(first-half destination registers, second-half source registers)] of every reduction in the assembly"""
    out, first, second = [], [], []
    for line in asm.splitlines():
        m = DPP.match(line)
        if m and m.group(4) == "0x3":
            if second:           # a new instance starts
                out.append((first, second))
                first, second = [], []
            assert m.group(1) == m.group(2) == m.group(3), line
            first.append(m.group(1))
        elif m and m.group(4) == "0xc":
            assert m.group(2) == m.group(3), line
            second.append((m.group(1), m.group(2)))
        elif first and second and not m and line.strip().startswith("v_add_f32_dpp") and "row_shl:4" in line:
            out.append((first, second))
            first, second = [], []
    if first and second:
        out.append((first, second))
    return out


def check(asm=None):
    inst = instances(asm if asm is not None else assembly())
    assert inst, "no pair_row_reduce instance found in the assembly (did the kernel change?)"
    report = []
    for first, second in inst:
        assert len(first) == len(second), (first, second)   # (9 / 12: pair_row_reduce9 / 12; 3 / 4: the same scheme in the smaller joint reductions)
        assert [d for d, _ in second] == first, "second half does not write the a[] registers in order: %r %r" % (first, second)
        srcs = [s for _, s in second]
        clash = sorted(set(first) & set(srcs))
        assert not clash, "pair_row_reduce%d: b[] input(s) %s share a VGPR with an a[] register that has already been overwritten" % (len(first), clash)
        report.append((len(first), first, srcs))
    sizes = {r[0] for r in report}
    assert 9 in sizes and 12 in sizes, "pair_row_reduce9 / pair_row_reduce12 not found in the assembly (sizes seen: %r)" % sorted(sizes)
    return report


if __name__ == "__main__":
    for ns, a, b in check():
        print("pair_row_reduce%d: a[] = %s; b[] = %s -- disjoint" % (ns, " ".join(a), " ".join(b)))
