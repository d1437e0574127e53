#!/bin/bash
# instruction-cache counters over ONE pass of the benchmarked path: bash tools/pmc_icache.sh <tag>   (through gpurun, from the repo root)
set -e
TAG=${1:-ic}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_icache_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail > $OUT/avail.txt 2>&1 || true
grep -i -E "ICACHE|IFETCH|INST_CACHE" $OUT/avail.txt | head -40 > $OUT/avail_icache.txt || true
i=0
for grp in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH GRBM_GUI_ACTIVE" "SQC_ICACHE_MISSES_DUPLICATE SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE"; do
  i=$((i + 1))
  rocprofv3 --pmc $grp --output-format csv -d $OUT/p$i -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-profile --no-exact-range > $OUT/p$i.out 2> $OUT/p$i.err || { echo "pass $i FAILED"; tail -5 $OUT/p$i.err; }
  echo "pass $i done"
done
python3 $ROOT/tools/pmc_summary.py $OUT > $OUT/SUMMARY.md || true
find $OUT -name "*.csv" -size +5M -delete || true
