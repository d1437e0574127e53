#!/bin/bash
# session: k-major fragment layout -- correctness of every W-direct kernel, then A/B stagger, stamps
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out/s3
timeout -k 10 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_pipeline.py tests/test_gpu_flow.py -x -q > gpurun_out/s3/tests.log 2>&1 || { tail -30 gpurun_out/s3/tests.log; exit 1; }
tail -n 2 gpurun_out/s3/tests.log
B="--no-cpu-baseline --no-exact-range"
JV_NO_FF_STAGGER=1 python bench.py $B > gpurun_out/s3/bench_lock.json 2> gpurun_out/s3/err.log || tail -n 5 gpurun_out/s3/err.log
python bench.py $B > gpurun_out/s3/bench_stag.json 2> gpurun_out/s3/err.log || tail -n 5 gpurun_out/s3/err.log
JV_NO_FF_STAGGER=1 python bench.py $B > gpurun_out/s3/bench_lock2.json 2> gpurun_out/s3/err.log || tail -n 5 gpurun_out/s3/err.log
python - <<'PY'
import json
for n in ("lock", "stag", "lock2"):
    j = json.loads([l for l in open(f"gpurun_out/s3/bench_{n}.json") if l.startswith("{")][-1])
    ks = j["kernels"]
    print(n, j["ms_per_step"], {k: round(1e3 * v["ms_per_step"] / v["launches"], 2) for k, v in ks.items() if k.startswith(("rowblock", "rowconv", "rowgemm", "hiftconv", "attn"))})
PY
JV_NO_FF_STAGGER=1 bash tools/rb_ablate.sh lock 0 > /dev/null
cat gpurun_out/rb_ablate_lock.txt
