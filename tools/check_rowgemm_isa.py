#!/usr/bin/env python3
"""Build-time check of the hand-managed register double buffer of the W-direct kernels: rowgemm_wd_kernel, rowgemm_wa_kernel,
rowffn_kernel (rowgemm_kernel.h), rowconv_wd_kernel (rowconv_kernel.h), rowblock_kernel (rowblock_kernel.h) hiftconv_kernel and hiftpair_kernel
(hiftconv_kernel.h: the vocoder's 72 ResBlock convolutions).

The weight fragments are loaded by inline asm (global_load_dwordx4) and waited for by a counted s_waitcnt the compiler does
not know about.  That is only sound if, in the generated code, the destination registers of those loads are touched by
nothing except (a) the asm loads, (b) v_mfma instructions reading them as an operand, (c) code ahead of the register's
first load (the straight-line prologue, where nothing is in flight in it yet).  A register-allocator copy or spill of one of them while a load is in flight would read stale data without
any tool noticing.  This script compiles rowgemm.hip, rowblock.hip and hiftconv.hip to assembly and asserts exactly that for every instantiation.

Second check, attention_s.hip (attn64_s_kernel): that kernel keeps its O accumulators and high Q planes in AGPRs it names
itself inside inline asm (a32 and up), without telling the compiler.  Sound only if the compiler-generated code of the kernel
touches no AGPR from a32 up and spills nothing (a scratch reload would not be a hazard, but the kernel is written to fit): asserted here.

usage: python tools/check_rowgemm_isa.py   (exit code 0 = clean)"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRCS = [os.path.join(ROOT, "jyutvoice_amd", "csrc", f) for f in ("rowgemm.hip", "rowblock.hip", "hiftconv.hip")]
CLANG = "/opt/rocm/lib/llvm/bin/clang++"
ATTN_S = os.path.join(ROOT, "jyutvoice_amd", "csrc", "attention_s.hip")
sys.path.insert(0, ROOT)
from jyutvoice_amd.build import FILE_FLAGS, FLAGS as LIB_FLAGS      # exactly the flags the library is built with (-fPIC, JV_EXTRA_FLAGS, ...)

FLAGS0 = list(LIB_FLAGS) + ["-x", "hip", "--cuda-device-only", "-S"]

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
        if in_asm and t.startswith("global_load_dwordx4") and "window rows" not in t:      # (rowconv's A-window loads are asm too:
            #  ordinary values once landed_a() has passed, read by the staging code -- not part of the fragment buffer)
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


AGPR = re.compile(r"(?<![\w.])a\[?(\d+)(?::(\d+))?")
AGPR_OWN = 32      # attention_s.hip leaves a[0:31] to the compiler


def check_agpr_owner(name, body):
    """no AGPR operand from a32 up and no scratch access outside inline asm"""
    bad = []
    in_asm = False
    n_asm = 0
    for i, ln in enumerate(body.split("\n")):
        t = ln.strip()
        if t.startswith(";;#ASMSTART"):
            in_asm = True
            n_asm += 1
            continue
        if t.startswith(";;#ASMEND"):
            in_asm = False
            continue
        t = t.split(";")[0].strip()
        if in_asm or not t or t.endswith(":") or t.startswith("."):
            continue
        parts = t.split(None, 1)
        if parts[0].startswith("scratch_"):
            bad.append((i, ln))
        elif len(parts) > 1:
            for m in AGPR.finditer(parts[1]):
                if max(int(m.group(1)), int(m.group(2) or 0)) >= AGPR_OWN:
                    bad.append((i, ln))
    return n_asm, bad


def main():
    s = ""
    with tempfile.TemporaryDirectory() as d:
        for src, extra in [(f, []) for f in SRCS + [ATTN_S]] + [(ATTN_S, ["-DJV_TUNING"])]:      # (the tuning build's variants too)
            out = os.path.join(d, os.path.basename(src) + (".tune" if extra else "") + ".s")
            FLAGS = FLAGS0 + FILE_FLAGS.get(os.path.basename(src), []) + extra
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
    for m in re.finditer(r"^(_ZN2jv\d+(rowgemm_wd_kernel|rowgemm_wa_kernel|rowffn_kernel|rowconv_wd_kernel|rowblock_kernel|rowres_kernel|hiftconv_kernel|hiftpair_kernel)\w+):\s*;.*?\n(.*?)\.end_amdhsa_kernel", s, re.S | re.M):
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
    for m in re.finditer(r"^(_ZN2jv\w*attn64_s_kernel\w+):\s*;.*?\n(.*?)\.end_amdhsa_kernel", s, re.S | re.M):
        name, body = m.group(1), m.group(2).split("s_endpgm")[0]
        n_asm, bad = check_agpr_owner(name, body)
        n_k += 1
        if n_asm == 0:
            print("FAIL", name, ": no inline asm found")
            n_bad += 1
        for i, ln in bad:
            print("FAIL", name, "line", i, ": compiler-generated AGPR / scratch use:", ln.strip())
            n_bad += 1
    print(f"checked {n_k} kernels, {n_bad} violations")
    return 1 if (n_bad or n_k == 0) else 0


if __name__ == "__main__":
    sys.exit(main())
