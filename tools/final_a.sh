#!/bin/bash
# round artefacts, call A: the whole GPU suite on the final build, then the headline bench line
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out/final
(time timeout -k 10 1000 python -m pytest tests -m gpu -x -q) > gpurun_out/final/gputests.log 2>&1 || { tail -n 40 gpurun_out/final/gputests.log; exit 1; }
tail -n 5 gpurun_out/final/gputests.log
python bench.py > gpurun_out/final/bench.json 2> gpurun_out/final/bench.err || { tail -n 5 gpurun_out/final/bench.err; exit 1; }
tail -c 1500 gpurun_out/final/bench.json
