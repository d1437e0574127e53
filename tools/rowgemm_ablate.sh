#!/bin/bash
# main-loop ablations of the row-owning GEMM (tuning build: JV_TUNING=1 JV_BUILD_TAG=tune python -m jyutvoice_amd.build):
#   bash tools/rowgemm_ablate.sh "0 1 2 3 4 5"      (JV_RG_ABLATE bits: 1 no DMA in the loop, 2 no LDS reads + MFMAs, 4 no waits/barriers)
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/rgablate
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export JYUTVOICE_HIP_LIB=$ROOT/jyutvoice_amd/libjyutvoice_hip.tune.so JV_OP_ROWGEMM=1 JV_ROWGEMM_EPI=${EPI:-plain}
for ab in ${1:-0 1 2 4}; do
  JV_RG_ABLATE=$ab rocprofv3 --kernel-trace --output-format csv -d $OUT/a$ab -- python3 $ROOT/tools/gemm_bench.py > $OUT/a$ab.out 2> $OUT/a$ab.err || { tail -5 $OUT/a$ab.err; exit 1; }
  python3 - <<PY
import csv, glob, os
names = ["qkv K256 N1536", "ff1 K256 N1024", "ff2 K1024 N256", "out K512 N256", "res K256 N256"]
for f in glob.glob(os.path.join("$OUT", "a$ab", "**", "*kernel_trace.csv"), recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if "rowgemm_" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    per = len(rows) // len(names)
    out = []
    for i, nm in enumerate(names):
        us = sorted((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows[i * per:(i + 1) * per][2:])
        out.append(f"{nm.split()[0]} {us[len(us)//2]:6.1f}")
    print("ablate $ab: " + "  ".join(out))
PY
  rm -rf $OUT/a$ab
done
