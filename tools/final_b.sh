#!/bin/bash
# round artefacts, call B: rocprofv3 kernel trace + PMC traffic of the headline (C3) and of BASELINE.json configs[1] (C2)
cd ${GRAFT_REPO_ROOT:-$(pwd)}
bash tools/profile.sh r04 > gpurun_out/prof_r04.log 2>&1 || { tail -n 20 gpurun_out/prof_r04.log; exit 1; }
tail -n 3 gpurun_out/prof_r04.log
JV_PROFILE_ARGS="--workload c2 --batch 8 --tokens 256" bash tools/profile.sh r04c2 > gpurun_out/prof_r04c2.log 2>&1 || { tail -n 20 gpurun_out/prof_r04c2.log; exit 1; }
tail -n 3 gpurun_out/prof_r04c2.log
