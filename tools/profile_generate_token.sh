#!/bin/bash
# Kernel timeline of ONE greedy token of the SHIPPED path - AKI.generate on the one-launch decode chain: chain, head GEMV,
# aki_greedy_pick_embed (pick + the next token's embedding row), issued eagerly - on the GPU box (run through gpurun):
#   bash tools/profile_generate_token.sh <tag>  -> gpurun_out/<tag>_decode_token_trace.txt, gpurun_out/<tag>_decode_kernels.txt
set -u
TAG=${1:-r05}
OUT=gpurun_out/prof_generate_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/generate_bench.py --new 48 --rounds 1 > $OUT/bench.log 2>&1
tail -1 $OUT/bench.log | cut -c1-400
python3 tools/decode_trace_summary.py $OUT/trace gpurun_out/${TAG}_decode
