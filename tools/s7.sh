#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out/s7
B="--no-cpu-baseline --no-exact-range"
for v in base u8 base2 u8b; do
  case $v in u8*) export JYUTVOICE_HIP_LIB=$PWD/jyutvoice_amd/libjyutvoice_hip.u8.so;; *) unset JYUTVOICE_HIP_LIB;; esac
  python bench.py $B > gpurun_out/s7/bench_$v.json 2> gpurun_out/s7/err.log || tail -n 5 gpurun_out/s7/err.log
done
python - <<'PY'
import json
for n in ("base", "u8", "base2", "u8b"):
    j = json.loads([l for l in open(f"gpurun_out/s7/bench_{n}.json") if l.startswith("{")][-1])
    ks = j["kernels"]
    print(n, j["ms_per_step"], j["stage_ms"]["hift"], {k: round(1e3 * v["ms_per_step"] / v["launches"], 2) for k, v in ks.items() if k.startswith("hiftconv")})
PY
