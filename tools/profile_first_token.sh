#!/bin/bash
# Kernel trace of one first token of AKI.generate on the GPU box (run through gpurun):
#   bash tools/profile_first_token.sh <tag>   -> gpurun_out/<tag>_first_token_trace.txt (phases, kernels, idle time of ONE prefill)
set -u
TAG=${1:-r05}
OUT=gpurun_out/prof_first_token_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 tools/first_token_profile.py > $OUT/run.log 2>&1
tail -1 $OUT/run.log | cut -c1-400
python3 tools/first_token_profile.py --summarize $OUT/trace gpurun_out/${TAG}
