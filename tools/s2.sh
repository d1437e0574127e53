#!/bin/bash
# session: stagger correctness + A/B + ablations
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out/s2
timeout -k 10 600 python -m pytest tests/test_gpu_pipeline.py -x -q -k "stagger or fused_block or split_qkv or golden" > gpurun_out/s2/tests.log 2>&1 || { tail -20 gpurun_out/s2/tests.log; exit 1; }
tail -2 gpurun_out/s2/tests.log
B="--no-cpu-baseline --no-exact-range"
python bench.py $B > gpurun_out/s2/bench_stag.json 2> gpurun_out/s2/err.log || tail -5 gpurun_out/s2/err.log
JV_NO_FF_STAGGER=1 python bench.py $B > gpurun_out/s2/bench_lock.json 2> gpurun_out/s2/err.log || tail -5 gpurun_out/s2/err.log
python bench.py $B > gpurun_out/s2/bench_stag2.json 2> gpurun_out/s2/err.log || tail -5 gpurun_out/s2/err.log
python - <<'PY'
import json
for n in ("stag", "lock", "stag2"):
    j = json.loads([l for l in open(f"gpurun_out/s2/bench_{n}.json") if l.startswith("{")][-1])
    ks = j["kernels"]
    print(n, j["ms_per_step"], {k: round(1e3 * v["ms_per_step"] / v["launches"], 2) for k, v in ks.items() if k.startswith("rowblock")})
PY
bash tools/rb_ablate.sh stag 0 2 8 16 26 > /dev/null
JV_NO_FF_STAGGER=1 bash tools/rb_ablate.sh lock 0 1 2 4 8 16 26 > /dev/null
cat gpurun_out/rb_ablate_stag.txt gpurun_out/rb_ablate_lock.txt
