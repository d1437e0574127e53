#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out/s5
timeout -k 10 900 python -m pytest tests/test_gpu_ragged.py tests/test_gpu_pipeline.py tests/test_gpu_flow.py tests/test_gpu_dist.py tests/test_gpu_regimes.py -x -q > gpurun_out/s5/tests.log 2>&1 || { tail -n 40 gpurun_out/s5/tests.log; exit 1; }
tail -n 2 gpurun_out/s5/tests.log
B="--no-cpu-baseline --no-exact-range"
python bench.py $B --ragged > gpurun_out/s5/bench_ragged.json 2> gpurun_out/s5/err.log || tail -n 5 gpurun_out/s5/err.log
JV_NO_COMPACT=1 python bench.py $B --ragged > gpurun_out/s5/bench_ragged_uniform.json 2> gpurun_out/s5/err.log || tail -n 5 gpurun_out/s5/err.log
python bench.py $B > gpurun_out/s5/bench.json 2> gpurun_out/s5/err.log || tail -n 5 gpurun_out/s5/err.log
python - <<'PY'
import json
for n in ("bench_ragged", "bench_ragged_uniform", "bench"):
    j = json.loads([l for l in open(f"gpurun_out/s5/{n}.json") if l.startswith("{")][-1])
    ks = j["kernels"]
    print(n, j["value"], j["ms_per_step"], j.get("stage_ms", {}).get("cfm_loop"), {k: round(1e3 * v["ms_per_step"] / v["launches"], 2) for k, v in list(ks.items())[:6]})
PY
