#!/bin/bash
# round artefacts, call C: the default bench line (now with roofline.traffic from the committed PMC pass), the C2 line, the other configurations
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out/final
python bench.py > gpurun_out/final/bench2.json 2> gpurun_out/final/bench2.err || { tail -n 5 gpurun_out/final/bench2.err; exit 1; }
python bench.py --workload c2 --batch 8 --tokens 256 --no-exact-range > gpurun_out/final/bench_c2.json 2> gpurun_out/final/bench_c2.err || { tail -n 5 gpurun_out/final/bench_c2.err; exit 1; }
python tools/bench_configs.py r04
