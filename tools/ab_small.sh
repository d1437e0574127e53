#!/bin/bash
# same-box A/B of the per-solve timestep embeddings and the norm1 fold at 1 / 8 / 32 utterances (through gpurun, repo root):
#   bash tools/ab_small.sh      -> gpurun_out/abs_*.json, gpurun_out/t2.log
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
B="--no-cpu-baseline --no-exact-range --no-profile"
python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_seam.py -q -x -k "timestep_embeddings or ln_fold or split_qkv or seam or golden or streaming" > gpurun_out/t2.log 2>&1
python bench.py --batch 1 --tokens 64 $B > gpurun_out/abs_b1_new.json 2> gpurun_out/err.log
JV_NO_TEMB_PRE=1 JV_NO_LN_FOLD=1 python bench.py --batch 1 --tokens 64 $B > gpurun_out/abs_b1_old.json 2> gpurun_out/err.log
python bench.py --batch 8 $B > gpurun_out/abs_b8_new.json 2> gpurun_out/err.log
JV_NO_TEMB_PRE=1 python bench.py --batch 8 $B > gpurun_out/abs_b8_old.json 2> gpurun_out/err.log
python bench.py $B > gpurun_out/abs_b32_new.json 2> gpurun_out/err.log
JV_NO_TEMB_PRE=1 python bench.py $B > gpurun_out/abs_b32_old.json 2> gpurun_out/err.log
