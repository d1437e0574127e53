#!/bin/bash
# round artefacts, call E (after profiles/ holds this build's PMC pass): the driver's command, the C2 line, the regime sweep
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out/final
python bench.py > gpurun_out/final/bench2.json 2> gpurun_out/final/bench2.err || { tail -n 5 gpurun_out/final/bench2.err; exit 1; }
python bench.py --workload c2 --batch 8 --tokens 256 --no-exact-range > gpurun_out/final/bench_c2.json 2> gpurun_out/final/bench_c2.err || { tail -n 5 gpurun_out/final/bench_c2.err; exit 1; }
python tools/regime_sweep.py --timesteps 4 --batches 1,2,4,6,8,12,16,24,32,48,64 > gpurun_out/final/regime_sweep.txt 2> gpurun_out/final/regime_sweep.err || tail -n 5 gpurun_out/final/regime_sweep.err
tail -n 16 gpurun_out/final/regime_sweep.txt
