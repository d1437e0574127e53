#!/bin/bash
# where a fused transformer-block launch spends its time: ablations of rowblock_kernel on the C3 shape (tuning build, through
# gpurun, repo root; results of the ablated runs are wrong by design -- only their launch times are read):
#   JV_TUNING=1 JV_BUILD_TAG=tune python -m jyutvoice_amd.build   (here, before the call)
#   bash tools/rb_ablate.sh [tag] [ablate values...]      -> gpurun_out/rb_ablate_<tag>.txt
# bits (RowBlockArgs::ablate): 1 GELU pass without arithmetic, 2 no GELU pass, 4 phase C without global stores, 8 phase C
# without chunk epilogues, 16 residual epilogues without row passes
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-0}
shift || true
VALS=${@:-0 1 2 4 8 16 26}
LIB=${JV_LIB:-$PWD/jyutvoice_amd/libjyutvoice_hip.tune.so}
OUT=gpurun_out/rb_ablate_$TAG.txt
: > $OUT
B="--workload c2 --batch ${JV_BATCH:-32} --tokens 150 --timesteps 2 --steps 2 --warmup 1 --no-cpu-baseline --no-exact-range"
for v in $VALS; do
  JYUTVOICE_HIP_LIB=$LIB JV_RB_ABLATE=$v python bench.py $B > gpurun_out/rb_ab.json 2> gpurun_out/rb_ab.err || { tail -5 gpurun_out/rb_ab.err; }
  python - $v >> $OUT <<'PY'
import json, sys
v = sys.argv[1]
try:
    j = json.loads([l for l in open("gpurun_out/rb_ab.json") if l.startswith("{")][-1])
    ks = {k: d for k, d in j.get("kernels", {}).items() if k.startswith("rowblock")}
    print(f"ablate {v:>4s}: ms_per_step {j['ms_per_step']:8.3f}  " + "  ".join(f"{k} {1e3 * d['ms_per_step'] * j.get('profiled_steps', 1) / d['launches']:.2f} us x{d['launches']}" for k, d in ks.items()))
except Exception as e:
    print(f"ablate {v}: failed ({e})")
PY
done
JYUTVOICE_HIP_LIB=$LIB JV_RB_STAMPS=1 python bench.py $B --steps 1 --no-profile > gpurun_out/rb_ab.json 2> gpurun_out/rb_stamps.err || true
grep "stamps" gpurun_out/rb_stamps.err | tail -3 >> $OUT
cat $OUT
