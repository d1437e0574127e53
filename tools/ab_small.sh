#!/bin/bash
# same-box A/B of the small-batch changes at 1 / 2 utterances (through gpurun, repo root):
#   bash tools/ab_small.sh      -> gpurun_out/abs_*.json
# new = default; old = JV_NO_TEMB_PRE=1 JV_NO_LN_FOLD=1 (the launch list of the round's first session)
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
B="--no-cpu-baseline --no-exact-range --no-profile"
python bench.py --batch 1 --tokens 64 $B > gpurun_out/abs_b1_new.json 2> gpurun_out/err.log
JV_NO_TEMB_PRE=1 JV_NO_LN_FOLD=1 python bench.py --batch 1 --tokens 64 $B > gpurun_out/abs_b1_old.json 2> gpurun_out/err.log
python bench.py --batch 2 $B > gpurun_out/abs_b2_new.json 2> gpurun_out/err.log
JV_NO_TEMB_PRE=1 JV_NO_LN_FOLD=1 python bench.py --batch 2 $B > gpurun_out/abs_b2_old.json 2> gpurun_out/err.log
