#!/bin/bash
# kernel durations of the row-owning GEMM against the tile kernels on the estimator's linears (run through gpurun):
#   bash tools/rowgemm_prof.sh <tag>
set -e
TAG=${1:-rg}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/rgprof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for epi in plain gelu res res_ln; do
  JV_OP_ROWGEMM=1 JV_ROWGEMM_EPI=$epi JV_ONLY=${JV_ONLY:-} rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$epi -- python3 $ROOT/tools/gemm_bench.py > $OUT/$epi.out 2> $OUT/$epi.err || { tail -5 $OUT/$epi.err; exit 1; }
done
JV_OP_H3=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tile -- python3 $ROOT/tools/gemm_bench.py > $OUT/tile.out 2> $OUT/tile.err || { tail -5 $OUT/tile.err; exit 1; }
python3 - <<PY
import csv, glob, os
names = ["qkv K256 N1536", "ff1 K256 N1024", "ff2 K1024 N256", "out K512 N256", "res K256 N256"]
for d in ("plain", "gelu", "res", "res_ln", "tile"):
    for f in glob.glob(os.path.join("$OUT", d, "**", "*kernel_trace.csv"), recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if "rowgemm_" in r["Kernel_Name"] or "conv_gemm_x6_kernel" in r["Kernel_Name"]]
        rows.sort(key=lambda r: int(r["Start_Timestamp"]))
        sh = names if d in ("plain", "gelu", "tile") else names[2:]
        per = len(rows) // len(sh) if sh else 0
        for i, nm in enumerate(sh):
            grp = rows[i * per:(i + 1) * per][2:]
            if not grp:
                continue
            us = sorted((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in grp)
            print(f"{d:7s} {nm:16s} {grp[0]['Kernel_Name'][9:60]:52s} n={len(us):3d} median {us[len(us)//2]:7.1f} us  min {us[0]:7.1f}")
PY
find $OUT -name "*kernel_trace.csv" -delete || true
