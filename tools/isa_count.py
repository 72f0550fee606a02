#!/usr/bin/env python3
"""Instruction-level count of the BUILT inner loop of the blend kernels (gfx950 assembly of csrc/blend_bwd.hip / blend_fwd.hip as
build.sh compiles them; no GPU needed): per basic block of the innermost loop that holds the exp (one trip = one PAIR of list
entries), how many VALU (plain / packed / DPP / transcendental), LDS, scalar, branch and memory instructions -- and which block is
the both-entries-alive path (the one that holds the joint DPP reduction).  usage: python tools/isa_count.py [blend_bwd|blend_fwd]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def assembly(tu):
    src = os.path.join(ROOT, "4d-gaussian-splatting_amd", "csrc", tu + ".hip")
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "--cuda-device-only", "-S", "-o", "-", src, "-I", os.path.join(ROOT, "include")]
    return subprocess.run(cmd, check=True, capture_output=True, text=True).stdout


def classify(op):
    if op.startswith("v_"):
        if op.startswith(("v_exp", "v_rcp", "v_log", "v_sqrt", "v_rsq", "v_sin", "v_cos")):
            return "valu_trans"
        if op.startswith("v_pk_"):
            return "valu_packed"
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op.startswith(("s_waitcnt", "s_nop", "s_barrier")):
        return "wait"
    if op.startswith("s_"):
        return "salu"
    return "other"


def kernels(asm):
    """name -> list of lines of every kernel in the assembly"""
    out, cur, name = {}, None, None
    for line in asm.splitlines():
        m = re.match(r"^(_ZN4fdgs\w+):", line)
        if m:
            name, cur = m.group(1), []
            out[name] = cur
        elif line.startswith(".Lfunc_end"):
            cur = None
        elif cur is not None:
            cur.append(line)
    return out


def inner_loop(lines):
    """(first, last) line index of the innermost loop (by the compiler's own loop comments) that contains an exp"""
    hdr = [i for i, l in enumerate(lines) if "Inner Loop Header" in l]
    best = None
    for h in hdr:
        # the header label is the line before the comment; the loop ends at the last line that says "in Loop: Header=<label>" ...
        lab = re.match(r"^(\.LBB\d+_\d+):", lines[h - 1])
        if not lab:
            continue
        tag = "Header=" + lab.group(1)[2:]
        members = [i for i, l in enumerate(lines) if tag in l]
        end = (max(members) if members else h)
        # ... plus that block's body up to the next label
        j = end + 1
        while j < len(lines) and not re.match(r"^\.LBB\d+_\d+:", lines[j]):
            j += 1
        body = lines[h - 1:j]
        if any("v_exp_f32" in l for l in body):
            best = (h - 1, j)
    return best


def count(lines, lo, hi):
    new = lambda label: {"label": label, "n": {}, "dpp": 0, "reduce": 0}   # noqa: E731
    blocks, cur, part = [], new("(loop header)"), 0
    for l in lines[lo:hi]:
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            if sum(cur["n"].values()):
                blocks.append(cur)
            cur, part = new(m.group(1)), 0
            continue
        t = l.strip().split()
        if not t or t[0].startswith((";", ".", "//")):
            continue
        c = classify(t[0])
        cur["n"][c] = cur["n"].get(c, 0) + 1
        if "_dpp" in t[0] or "row_" in l or "quad_perm" in l:
            cur["dpp"] += 1
        if "row_ror:8" in l and "bank_mask:0xc" in l:
            cur["reduce"] += 1
        if c == "branch":      # a basic block ends at a branch as well as at a label
            base = cur["label"].split(" +")[0]
            blocks.append(cur)
            part += 1
            cur = new("%s +%d" % (base, part))
    if sum(cur["n"].values()):
        blocks.append(cur)
    return blocks


if __name__ == "__main__":
    tu = sys.argv[1] if len(sys.argv) > 1 else "blend_bwd"
    for name, lines in kernels(assembly(tu)).items():
        if "blend" not in name or "debug" in name:
            continue
        span = inner_loop(lines)
        if not span:
            continue
        print(name)
        tot = {}
        for b in count(lines, *span):
            v = sum(b["n"].get(k, 0) for k in ("valu", "valu_packed", "valu_trans"))
            print("  %-16s VALU %3d (packed %2d, transcendental %d, DPP %2d)  LDS %2d  SALU %2d  branch %d  vmem %d  wait %2d%s" % (
                b["label"], v, b["n"].get("valu_packed", 0), b["n"].get("valu_trans", 0), b["dpp"], b["n"].get("lds", 0), b["n"].get("salu", 0),
                b["n"].get("branch", 0), b["n"].get("vmem", 0), b["n"].get("wait", 0),
                "   <- both entries alive: joint DPP reduction of %d slots" % b["reduce"] if b["reduce"] in (9, 12) else ("   <- one entry alive" if b["dpp"] > 15 else "")))
            for k, x in b["n"].items():
                tot[k] = tot.get(k, 0) + x
        print("  loop total  ", tot)
