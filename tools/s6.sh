#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out/s6
timeout -k 10 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_flow.py tests/test_gpu_ragged.py tests/test_gpu_ops.py -x -q > gpurun_out/s6/tests.log 2>&1 || { tail -n 40 gpurun_out/s6/tests.log; exit 1; }
tail -n 2 gpurun_out/s6/tests.log
B="--no-cpu-baseline --no-exact-range"
python bench.py $B > gpurun_out/s6/bench_fold.json 2> gpurun_out/s6/err.log || tail -n 5 gpurun_out/s6/err.log
JV_NO_RES_FOLD=1 python bench.py $B > gpurun_out/s6/bench_nofold.json 2> gpurun_out/s6/err.log || tail -n 5 gpurun_out/s6/err.log
python bench.py $B > gpurun_out/s6/bench_fold2.json 2> gpurun_out/s6/err.log || tail -n 5 gpurun_out/s6/err.log
python - <<'PY'
import json
for n in ("bench_fold", "bench_nofold", "bench_fold2"):
    j = json.loads([l for l in open(f"gpurun_out/s6/{n}.json") if l.startswith("{")][-1])
    ks = j["kernels"]
    print(n, j["value"], j["ms_per_step"], j.get("conv_stack", {}).get("ms_per_pass"), j.get("conv_stack", {}).get("alg_tflops"), {k: (v["launches"], round(1e3 * v["ms_per_step"] / v["launches"], 2)) for k, v in ks.items() if k.startswith(("rowconv", "conv_gemm_h3<64x64>"))})
PY
