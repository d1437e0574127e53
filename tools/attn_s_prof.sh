# kernel-trace durations of the attention kernels in the microbench (tools/gemm_bench.py), per key length and timing experiment
# usage (GPU box): bash tools/attn_s_prof.sh   -- needs the tuning build (JV_TUNING=1 JV_BUILD_TAG=tune python -m jyutvoice_amd.build)
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export JYUTVOICE_HIP_LIB=$ROOT/jyutvoice_amd/libjyutvoice_hip.tune.so
cd /tmp && export TMPDIR=/tmp
run() {      # tag, env...
  tag=$1; shift
  env JV_ONLY=attn JV_ATTN_L=300 JV_OP_ATTN_PL=1 "$@" rocprofv3 --kernel-trace --output-format csv -d /tmp/asp_$tag -- python3 $ROOT/tools/gemm_bench.py > /tmp/asp_$tag.out 2>/tmp/asp_$tag.err || { tail -5 /tmp/asp_$tag.err; exit 1; }
  python3 - <<PY
import csv, glob
for f in glob.glob("/tmp/asp_$tag/**/*kernel_trace.csv", recursive=True):
    rows=[r for r in csv.DictReader(open(f)) if "attn64" in r["Kernel_Name"]]
    us=sorted((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3 for r in rows[3:])
    print("%-28s %-40s n=%d median %.1f us  min %.1f" % ("$tag", rows[0]["Kernel_Name"][:40], len(us), us[len(us)//2], us[0]))
PY
}
for len in ${LENS:-32 300}; do
  run pl_$len JV_ATTN_LEN=$len
  run s_$len JV_ATTN_LEN=$len JV_OP_ATTN_SINGLE=1
  for e in ${EXPERS:-1 2}; do run s_exper${e}_$len JV_ATTN_LEN=$len JV_OP_ATTN_SINGLE=1 JV_AS_EXPER=$e; done
done
