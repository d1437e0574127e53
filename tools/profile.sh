#!/bin/bash
# Profiling recipe for the GPU box (run through gpurun from the repo root):
#   bash tools/profile.sh <tag>
# 1. rocprofv3 --kernel-trace --stats of the default bench (per-kernel time)
# 2. separate --pmc passes (FETCH_SIZE, WRITE_SIZE) on a 1-step run, as MI355X_MICROARCH.md prescribes
# JV_PROFILE_ARGS: extra bench.py arguments, e.g. "--workload c2 --batch 8 --tokens 256" (BASELINE.json configs[1])
set -e
X=${JV_PROFILE_ARGS:-}
TAG=${1:-r1}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-exact-range $X > $OUT/bench_under_rocprof.json 2> $OUT/trace.err || { tail -20 $OUT/trace.err; exit 1; }
echo "trace done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-profile --no-exact-range $X > $OUT/pmc_fetch.json 2> $OUT/pmc_fetch.err || { tail -20 $OUT/pmc_fetch.err; exit 1; }
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-profile --no-exact-range $X > $OUT/pmc_write.json 2> $OUT/pmc_write.err || { tail -20 $OUT/pmc_write.err; exit 1; }
echo "write done"
python3 $ROOT/tools/profile_summary.py $OUT > $OUT/SUMMARY.md
cat $OUT/SUMMARY.md
# keep only the small files (the raw per-dispatch CSVs are tens of MB)
find $OUT -name "*kernel_trace.csv" -size +20M -delete || true
