#!/bin/bash
# same-box A/B of the mid-batch q|k|v column split and the norm1 fold (run through gpurun from the repo root):
#   bash tools/ab_qkv_split.sh      -> gpurun_out/ab_*.json, gpurun_out/t1.log
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
B="--no-cpu-baseline --no-exact-range --no-profile"
python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_regimes.py tests/test_gpu_dist.py -q -x -k "split_qkv or fused_block or ln_fold or regime or dist" > gpurun_out/t1.log 2>&1
for b in 8 4 6 10 12; do
  python bench.py --batch $b $B > gpurun_out/ab_b${b}_new.json 2> gpurun_out/err.log
  JV_NO_QKV_SPLIT=1 JV_NO_LN_FOLD=1 python bench.py --batch $b $B > gpurun_out/ab_b${b}_old.json 2> gpurun_out/err.log
done
python bench.py --batch 8 --no-cpu-baseline --no-exact-range > gpurun_out/ab_b8_prof.json 2> gpurun_out/err.log
