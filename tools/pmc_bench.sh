#!/bin/bash
# SQ counter passes over ONE pass of the benchmarked path (bench.py --steps 1), every kernel of the pipeline at its real shapes:
#   bash tools/pmc_bench.sh <tag>      (through gpurun, from the repo root)
# Each pass is its own rocprofv3 --pmc run (no trace domains), as MI355X_MICROARCH.md prescribes; the program follows "--" directly.
set -e
TAG=${1:-pmcb}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_bench_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_MFMA" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  i=$((i + 1))
  rocprofv3 --pmc $grp --output-format csv -d $OUT/p$i -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-profile --no-exact-range > $OUT/p$i.out 2> $OUT/p$i.err || { echo "pass $i FAILED"; tail -20 $OUT/p$i.err; exit 1; }
  echo "pass $i done"
done
python3 $ROOT/tools/pmc_summary.py $OUT > $OUT/SUMMARY.md
head -60 $OUT/SUMMARY.md
find $OUT -name "*.csv" -size +5M -delete || true
