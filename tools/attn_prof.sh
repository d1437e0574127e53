set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for nw in 2 4 8; do
  JV_ATTN_NW=$nw JV_ONLY=attn JV_OP_ATTN_PL=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ap$nw -- python3 $ROOT/tools/gemm_bench.py > /tmp/ap$nw.out 2>/tmp/ap$nw.err || { tail -5 /tmp/ap$nw.err; exit 1; }
  python3 - <<PY
import csv, glob
for f in glob.glob("/tmp/ap$nw/**/*kernel_trace.csv", recursive=True):
    rows=[r for r in csv.DictReader(open(f)) if "attn64_pl" in r["Kernel_Name"]]
    rows.sort(key=lambda r:int(r["Start_Timestamp"]))
    per=len(rows)//2
    for i,nm in enumerate(("L=300 B'=64","L=512 B'=16")):
        us=sorted((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3 for r in rows[i*per:(i+1)*per][2:])
        print("NW=$nw", nm, "median %.1f us"%us[len(us)//2])
PY
done
