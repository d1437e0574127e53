#!/bin/bash
# per-kernel time of the default bench under rocprofv3 (run through gpurun from the repo root):
#   bash tools/trace.sh <tag> [bench.py flags...]       -> gpurun_out/trace_<tag>/SUMMARY.txt
set -e
TAG=${1:-t}
shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/trace_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile --no-exact-range "$@" > $OUT/bench.json 2> $OUT/trace.err || { tail -20 $OUT/trace.err; exit 1; }
python3 - <<PY > $OUT/SUMMARY.txt
import csv, glob, json, os
f = glob.glob(os.path.join("$OUT", "trace", "**", "*kernel_stats.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
b = json.load(open("$OUT/bench.json"))
print(f"bench under rocprofv3: {b['ms_per_step']} ms per step; kernel time {tot / 4e6:.1f} ms per pass (4 passes traced)")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:28]:
    print(f"{r['Name'][:100]:100s} calls/pass {int(r['Calls']) / 4:7.1f} avg {float(r['AverageNs']) / 1e3:8.1f} us  per pass {float(r['TotalDurationNs']) / 4e6:7.2f} ms")
PY
cat $OUT/SUMMARY.txt
find $OUT -name "*kernel_trace.csv" -delete || true
