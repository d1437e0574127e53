#!/usr/bin/env python3
"""Build-time check of the hand-managed register double buffer of the W-direct kernels: rowgemm_wd_kernel, rowgemm_wa_kernel,
rowffn_kernel (rowgemm_kernel.h), rowconv_wd_kernel (rowconv_kernel.h) and rowblock_kernel (rowblock_kernel.h).

The weight fragments are loaded by inline asm (global_load_dwordx4) and waited for by a counted s_waitcnt the compiler does
not know about.  That is only sound if, in the generated code, the destination registers of those loads are touched by
nothing except (a) the asm loads, (b) v_mfma instructions reading them as an operand, (c) code ahead of the register's
first load (the straight-line prologue, where nothing is in flight in it yet).  A register-allocator copy or spill of one of them while a load is in flight would read stale data without
any tool noticing.  This script compiles rowgemm.hip and rowblock.hip to assembly and asserts exactly that for every instantiation.

usage: python tools/check_rowgemm_isa.py   (exit code 0 = clean)"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRCS = [os.path.join(ROOT, "jyutvoice_amd", "csrc", f) for f in ("rowgemm.hip", "rowblock.hip")]
CLANG = "/opt/rocm/lib/llvm/bin/clang++"
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-gpu-rdc", "-fno-slp-vectorize", "-x", "hip", "--cuda-device-only", "-S"]

REG = re.compile(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b")


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1) is not None:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def check_kernel(name, body):
    lines = body.split("\n")
    in_asm = False
    loads = []          # (line index, dest regs)
    for i, ln in enumerate(lines):
        t = ln.strip()
        if t.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if t.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if in_asm and t.startswith("global_load_dwordx4"):
            dest = t.split(None, 1)[1].split(",")[0]
            loads.append((i, regs_of(dest)))
    if not loads:
        return 0, []
    buf = set().union(*[r for _, r in loads])
    first = {}          # register -> line of the first load into it: nothing is in flight in it before that
    for i, rs in loads:
        for r in rs:
            first.setdefault(r, i)
    bad = []
    in_asm = False
    for i, ln in enumerate(lines):
        t = ln.split(";")[0].strip() if not ln.strip().startswith(";;#") else ln.strip()
        if t.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if t.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not t or t.endswith(":") or t.startswith("."):
            continue
        if in_asm:
            continue
        parts = t.split(None, 1)
        if len(parts) < 2:
            continue
        mn, ops = parts
        opl = [o.strip() for o in ops.split(",")]
        touched = regs_of(ops) & buf
        if not touched:
            continue
        if mn.startswith("v_mfma"):
            if regs_of(opl[0]) & buf:
                bad.append((i, ln))      # an MFMA may read the buffer, never write it
            continue
        touched = {r for r in touched if i > first[r]}
        if not touched:
            continue                     # ahead of the register's first load (straight-line prologue): nothing in flight
        bad.append((i, ln))
    return len(loads), bad


def main():
    s = ""
    with tempfile.TemporaryDirectory() as d:
        for src in SRCS:
            out = os.path.join(d, os.path.basename(src) + ".s")
            r = subprocess.run([CLANG] + FLAGS + ["-o", out, src], capture_output=True, text=True)
            if r.returncode != 0:
                print(r.stderr[-3000:])
                return 2
            s += open(out).read()
            # -S does not run the assembler over the inline asm: an operand the assembler rejects only shows with -c
            r = subprocess.run([CLANG] + [f for f in FLAGS if f != "-S"] + ["-c", "-o", os.path.join(d, os.path.basename(src) + ".o"), src],
                               capture_output=True, text=True)
            if r.returncode != 0:
                print(r.stderr[-3000:])
                return 2
    n_k = n_bad = 0
    for m in re.finditer(r"^(_ZN2jv\d+(rowgemm_wd_kernel|rowgemm_wa_kernel|rowffn_kernel|rowconv_wd_kernel|rowblock_kernel)\w+):\s*;.*?\n(.*?)\.end_amdhsa_kernel", s, re.S | re.M):
        name, body = m.group(1), m.group(3)
        body = body.split("s_endpgm")[0]
        n_loads, bad = check_kernel(name, body)
        n_k += 1
        if n_loads == 0:
            print("FAIL", name, ": no asm loads found")
            n_bad += 1
        for i, ln in bad:
            print("FAIL", name, "line", i, ":", ln.strip())
            n_bad += 1
    print(f"checked {n_k} kernels, {n_bad} violations")
    return 1 if (n_bad or n_k == 0) else 0


if __name__ == "__main__":
    sys.exit(main())
