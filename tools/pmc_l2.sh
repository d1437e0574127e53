#!/bin/bash
# L2 request / hit / miss counters per kernel over one pass of a bench configuration (through gpurun, repo root):
#   bash tools/pmc_l2.sh <tag> [bench.py flags...]      -> gpurun_out/pmc_l2_<tag>/SUMMARY.txt
# One rocprofv3 --pmc run without trace domains, the program directly behind "--" (MI355X_MICROARCH.md).
set -e
TAG=${1:-l2}
shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_l2_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_READ_sum --output-format csv -d $OUT/p1 -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-profile --no-exact-range "$@" > $OUT/p1.out 2> $OUT/p1.err || { echo "pass FAILED"; tail -20 $OUT/p1.err; exit 1; }
python3 - <<PY > $OUT/SUMMARY.txt
import csv, glob, os, sys, collections
sys.path.insert(0, "$ROOT/tools")
f = glob.glob(os.path.join("$OUT", "p1", "**", "*counter_collection.csv"), recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"][:90]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[k].add(r["Dispatch_Id"])
print(f"{'kernel':90s} {'launches':>8s} {'L2 req / launch':>16s} {'MB at 128 B':>12s} {'hit rate':>9s} {'read share':>10s}")
for k, c in sorted(agg.items(), key=lambda kv: -kv[1].get("TCC_REQ_sum", 0))[:16]:
    n = len(cnt[k])
    req = c.get("TCC_REQ_sum", 0) / n
    hit, miss = c.get("TCC_HIT_sum", 0), c.get("TCC_MISS_sum", 0)
    print(f"{k:90s} {n:8d} {req:16.0f} {req * 128 / 1e6:12.1f} {hit / max(hit + miss, 1):9.3f} {c.get('TCC_READ_sum', 0) / max(c.get('TCC_REQ_sum', 1), 1):10.3f}")
PY
cat $OUT/SUMMARY.txt
find $OUT -name "*.csv" -size +5M -delete || true
