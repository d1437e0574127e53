#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out/s8
timeout -k 10 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_edges.py tests/test_gpu_ops.py -x -q -k "hift or synthesise or wav or vocoder or edge" > gpurun_out/s8/tests.log 2>&1 || { tail -n 40 gpurun_out/s8/tests.log; exit 1; }
tail -n 2 gpurun_out/s8/tests.log
B="--no-cpu-baseline --no-exact-range"
python bench.py $B > gpurun_out/s8/bench_pair.json 2> gpurun_out/s8/err.log || tail -n 5 gpurun_out/s8/err.log
JV_NO_HIFT_PAIR=1 python bench.py $B > gpurun_out/s8/bench_nopair.json 2> gpurun_out/s8/err.log || tail -n 5 gpurun_out/s8/err.log
python bench.py > gpurun_out/s8/bench_full.json 2> gpurun_out/s8/err_full.log || tail -n 5 gpurun_out/s8/err_full.log
python - <<'PY'
import json
for n in ("bench_pair", "bench_nopair", "bench_full"):
    j = json.loads([l for l in open(f"gpurun_out/s8/{n}.json") if l.startswith("{")][-1])
    ks = j["kernels"]
    print(n, j["value"], j["ms_per_step"], j["stage_ms"]["hift"], j.get("parity"), {k: (v["launches"], round(1e3 * v["ms_per_step"] / v["launches"], 2)) for k, v in ks.items() if k.startswith("hift")})
PY
