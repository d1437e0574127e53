"""Build libjyutvoice_hip.so in-tree with hipcc for gfx950 (no torch involved: the library is plain HIP/C++).

    python -m jyutvoice_amd.build [--force]

Objects are cached per source in jyutvoice_amd/csrc/build/ keyed by mtime of the source and headers.
"""
from __future__ import annotations

import glob
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
# JV_BUILD_TAG=x (tuning aid): objects in build/x/, library libjyutvoice_hip.x.so -- load it with JYUTVOICE_HIP_LIB=<path>;
# JV_EXTRA_FLAGS adds compiler flags to that variant, so two builds can be A/B'd inside one GPU session.
TAG = os.environ.get("JV_BUILD_TAG", "")
OBJ = os.path.join(CSRC, "build", TAG) if TAG else os.path.join(CSRC, "build")
LIB = os.path.join(HERE, f"libjyutvoice_hip.{TAG}.so" if TAG else "libjyutvoice_hip.so")
# -fno-slp-vectorize: packed f32 VALU (v_pk_add/mul/fma_f32) issues slower than the scalar pair beside MFMAs on gfx950
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wno-unused-result", "-fno-slp-vectorize"]
# per-file additions.  attention_s.hip owns the wave's 256 AGPRs by name inside its inline asm (O tiles and the high Q planes):
# the compiler must not park spilled VGPRs there (tools/check_rowgemm_isa.py checks the ISA for any AGPR use of its own)
FILE_FLAGS = {"attention_s.hip": ["-mllvm", "-amdgpu-spill-vgpr-to-agpr=0"]}
FLAGS += os.environ.get("JV_EXTRA_FLAGS", "").split()
if os.environ.get("JV_TUNING"):      # ablation switches + in-kernel stamps (tools/gemm_bench.py); use with --force
    FLAGS.append("-DJV_TUNING")


def source_hash() -> str:
    """sha256 over the kernel sources (csrc/*.hip, csrc/*.h, include/*.h; names and contents), first 16 hex digits: what a
    PMC traffic file under profiles/ is valid for (tools/profile_summary.py stores it, bench.py compares it)"""
    import hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.h")) +
                   glob.glob(os.path.join(HERE, "..", "include", "*.h")), key=os.path.basename)
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libjyutvoice_hip.so cannot be built (ROCm toolchain required)")


def build(force: bool = False, verbose: bool = True) -> str:
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    hdrs = glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(HERE, "..", "include", "*.h"))
    newest_hdr = max(os.path.getmtime(h) for h in hdrs)
    os.makedirs(OBJ, exist_ok=True)
    hipcc = _hipcc()
    jobs = []
    objs = []
    for s in srcs:
        o = os.path.join(OBJ, os.path.basename(s)[:-4] + ".o")
        objs.append(o)
        if force or not os.path.exists(o) or os.path.getmtime(o) < max(os.path.getmtime(s), newest_hdr):
            jobs.append((s, o))

    def cc(job):
        s, o = job
        r = subprocess.run([hipcc, *FLAGS, *FILE_FLAGS.get(os.path.basename(s), []), "-c", s, "-o", o], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {s}:\n{r.stdout}\n{r.stderr}")
        return s

    jobs.sort(key=lambda j: 0 if 'conv_gemm_x6_t' in j[0] else 1)      # the long compiles first
    if jobs:
        with ThreadPoolExecutor(max_workers=max(1, min(6, (os.cpu_count() or 4) - 2, len(jobs)))) as ex:
            for s in ex.map(cc, jobs):
                if verbose:
                    print(f"[jyutvoice_amd.build] compiled {os.path.basename(s)}", flush=True)
    if jobs or force or not os.path.exists(LIB):
        r = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
        if verbose:
            print(f"[jyutvoice_amd.build] linked {LIB}", flush=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
