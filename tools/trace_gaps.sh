#!/bin/bash
# Where a small batch's time goes: kernel durations against the idle time between consecutive kernels, from the
# rocprofv3 kernel trace of one bench configuration (run through gpurun from the repo root):
#   bash tools/trace_gaps.sh <tag> [bench.py flags...]       -> gpurun_out/gaps_<tag>/SUMMARY.txt
set -e
TAG=${1:-g}
shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/gaps_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile --no-exact-range "$@" > $OUT/bench.json 2> $OUT/trace.err || { tail -20 $OUT/trace.err; exit 1; }
python3 - <<PY > $OUT/SUMMARY.txt
import csv, glob, json, os, sys
sys.path.insert(0, "$ROOT/tools")
f = glob.glob(os.path.join("$OUT", "trace", "**", "*kernel_trace.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
b = json.load(open("$OUT/bench.json"))
# the last pass of the trace: the kernels behind the last occurrence of the pass's first kernel name sequence.  Simpler
# and robust: take the last len/4 kernels when 4 passes were traced (1 warm-up + 3 timed; the parity / stage passes
# that bench.py runs outside the timed region come first or are excluded by taking the tail)
names = [r["Kernel_Name"] for r in rows]
# one pass = the longest period d for which the trace holds two identical consecutive runs of d kernel names (the timed
# passes are identical launch sequences); analysed on the later of the two runs.  Fallback: the whole trace.
per, end = None, None
for e in range(len(names), 200, -1):
    last = names[e - 1]
    idx = [i for i in range(e - 1) if names[i] == last]
    for k in reversed(idx):
        d = e - 1 - k
        if d > 100 and e - 2 * d >= 0 and names[e - d:e] == names[e - 2 * d:e - d]:
            per, end = d, e
            break
    if per or len(names) - e > 3000:
        break
if per is None:
    per, end = len(rows), len(rows)
tail = rows[end - per:end]
t0, t1 = int(tail[0]["Start_Timestamp"]), int(tail[-1]["End_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in tail)
gaps = [min(int(tail[i + 1]["Start_Timestamp"]) - int(tail[i]["End_Timestamp"]), 200000) for i in range(len(tail) - 1)]
print(f"bench under rocprofv3: {b['ms_per_step']} ms per step; last pass: {per} kernels, span {(t1 - t0) / 1e6:.2f} ms, "
      f"busy {busy / 1e6:.2f} ms, idle between kernels {sum(max(g, 0) for g in gaps) / 1e6:.2f} ms "
      f"(median gap {sorted(gaps)[len(gaps) // 2] / 1e3:.2f} us, median kernel {sorted(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in tail)[per // 2] / 1e3:.2f} us)")
agg = {}
for i, r in enumerate(tail):
    n = r["Kernel_Name"]
    a = agg.setdefault(n, [0, 0, 0])
    a[0] += 1
    a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    if i + 1 < len(tail):
        a[2] += max(gaps[i], 0)
print(f"{'kernel':110s} {'calls':>6s} {'avg us':>8s} {'sum ms':>8s} {'gap after, avg us':>18s}")
for n, a in sorted(agg.items(), key=lambda kv: -(kv[1][1] + kv[1][2]))[:40]:
    print(f"{n[:110]:110s} {a[0]:6d} {a[1] / a[0] / 1e3:8.2f} {a[1] / 1e6:8.3f} {a[2] / a[0] / 1e3:18.2f}")
PY
cat $OUT/SUMMARY.txt
find $OUT -name "*kernel_trace.csv" -delete || true
