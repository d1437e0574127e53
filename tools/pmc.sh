#!/bin/bash
# SQ counter passes over tools/gemm_bench.py for one shape filter (run through gpurun from the repo root):
#   bash tools/pmc.sh <JV_ONLY filter: qkv|ff1|ff2|out|res|conv3|attn> <tag>
# Each pass is its own rocprofv3 --pmc run (no trace domains), as MI355X_MICROARCH.md prescribes.
set -e
ONLY=${1:-attn}
TAG=${2:-pmc}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_${TAG}_$ONLY
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export JV_ONLY=$ONLY JV_OP_X6=1      # add JV_OP_H3=1 in the environment for the fp16x3 forms
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_MFMA" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  i=$((i + 1))
  # a failed pass must not end up summarised as if it were complete (profile.sh exits the same way)
  rocprofv3 --pmc $grp --output-format csv -d $OUT/p$i -- python3 $ROOT/tools/gemm_bench.py > $OUT/p$i.out 2> $OUT/p$i.err || { echo "pass $i FAILED"; tail -20 $OUT/p$i.err; exit 1; }
  echo "pass $i done"
done
python3 $ROOT/tools/pmc_summary.py $OUT | tee $OUT/SUMMARY.md
find $OUT -name "*.csv" -size +5M -delete || true
