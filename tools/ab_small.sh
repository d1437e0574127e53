#!/bin/bash
# same-box A/B of the small-batch changes at 1 / 2 / 3 utterances (through gpurun, repo root):
#   bash tools/ab_small.sh      -> gpurun_out/abs_*.json, gpurun_out/t2.log
# new = default; old = JV_NO_ALLW=1 (two-buffer tile schedule), JV_NO_TEMB_PRE=1 JV_NO_LN_FOLD=1 (round 2's launch list)
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
B="--no-cpu-baseline --no-exact-range --no-profile"
python -m pytest tests/test_gpu_ops.py tests/test_gpu_pipeline.py tests/test_gpu_edges.py tests/test_gpu_flow.py -q -x > gpurun_out/t2.log 2>&1
python bench.py --batch 1 --tokens 64 $B > gpurun_out/abs_b1_new.json 2> gpurun_out/err.log
JV_DYNAMIC_ENV=1 JV_NO_ALLW=1 python bench.py --batch 1 --tokens 64 $B > gpurun_out/abs_b1_noallw.json 2> gpurun_out/err.log
JV_DYNAMIC_ENV=1 JV_NO_ALLW=1 JV_NO_TEMB_PRE=1 JV_NO_LN_FOLD=1 python bench.py --batch 1 --tokens 64 $B > gpurun_out/abs_b1_old.json 2> gpurun_out/err.log
python bench.py --batch 2 $B > gpurun_out/abs_b2_new.json 2> gpurun_out/err.log
JV_DYNAMIC_ENV=1 JV_NO_ALLW=1 python bench.py --batch 2 $B > gpurun_out/abs_b2_noallw.json 2> gpurun_out/err.log
python bench.py --batch 1 $B > gpurun_out/abs_b1x150_new.json 2> gpurun_out/err.log
JV_DYNAMIC_ENV=1 JV_NO_ALLW=1 python bench.py --batch 1 $B > gpurun_out/abs_b1x150_noallw.json 2> gpurun_out/err.log
bash tools/trace_gaps.sh b1n --batch 1 --tokens 64 > gpurun_out/gaps_b1n.log 2>&1
