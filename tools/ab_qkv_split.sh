#!/bin/bash
# same-box A/B of the mid-batch q|k|v column split and the norm1 fold (run through gpurun from the repo root):
#   bash tools/ab_qkv_split.sh      -> gpurun_out/ab_*.json, gpurun_out/t1.log, gpurun_out/full.log
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
B="--no-cpu-baseline --no-exact-range --no-profile"
python -m pytest tests/test_gpu_pipeline.py -q -x -k "split_qkv or fused_block or ln_fold" > gpurun_out/t1.log 2>&1
for b in 8 4 6; do
  python bench.py --batch $b $B > gpurun_out/ab_b${b}_new.json 2> gpurun_out/err.log
  JV_NO_QKV_SPLIT=1 JV_NO_LN_FOLD=1 python bench.py --batch $b $B > gpurun_out/ab_b${b}_old.json 2> gpurun_out/err.log
done
python bench.py $B > gpurun_out/ab_b32_new.json 2> gpurun_out/err.log
JV_NO_LN_FOLD=1 python bench.py $B > gpurun_out/ab_b32_old.json 2> gpurun_out/err.log
python bench.py --batch 8 --no-cpu-baseline --no-exact-range > gpurun_out/ab_b8_prof.json 2> gpurun_out/err.log
python -m pytest tests -m gpu -x -q > gpurun_out/full.log 2>&1
