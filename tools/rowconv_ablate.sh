#!/bin/bash
# main-loop ablations of the row-owning causal conv (tuning build): bash tools/rowconv_ablate.sh "0 1 2 8 16"
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
export JYUTVOICE_HIP_LIB=$ROOT/jyutvoice_amd/libjyutvoice_hip.tune.so JV_OP_ROWCONV=1 JV_ONLY=conv3
for ab in ${1:-0 1 2 8 16}; do
  JV_RG_ABLATE=$ab rocprofv3 --kernel-trace --output-format csv -d /tmp/rca$ab -- python3 $ROOT/tools/gemm_bench.py > /tmp/rca$ab.out 2> /tmp/rca$ab.err || { tail -5 /tmp/rca$ab.err; exit 1; }
  python3 - <<PY
import csv, glob
for f in glob.glob("/tmp/rca$ab/**/*kernel_trace.csv", recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if "rowconv_kernel" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    per = len(rows) // 3
    out = []
    for i, nm in enumerate(("conv3 K768", "conv3+LN K768", "conv3+LN K960")):
        us = sorted((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows[i * per:(i + 1) * per][1:])
        out.append(f"{nm} {us[len(us)//2]:6.1f}")
    print("ablate $ab: " + "  ".join(out))
PY
done
