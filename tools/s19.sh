set -e
mkdir -p gpurun_out/s19
timeout -k 10 300 python -m pytest tests/test_gpu_pipeline.py -x -q -m gpu -k "res_qkv or res_pair" > gpurun_out/s19/tests.log 2>&1
timeout -k 10 200 python bench.py --no-cpu-baseline --no-exact-range > gpurun_out/s19/bench_fused.json 2> gpurun_out/s19/bench_fused.err
JV_NO_RES_QKV=1 timeout -k 10 200 python bench.py --no-cpu-baseline --no-exact-range > gpurun_out/s19/bench_sep.json 2> gpurun_out/s19/bench_sep.err
timeout -k 10 200 python bench.py --no-cpu-baseline --no-exact-range > gpurun_out/s19/bench_fused2.json 2> gpurun_out/s19/bench_fused2.err
JV_NO_RES_QKV=1 timeout -k 10 200 python bench.py --no-cpu-baseline --no-exact-range > gpurun_out/s19/bench_sep2.json 2> gpurun_out/s19/bench_sep2.err
