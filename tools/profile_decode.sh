#!/bin/bash
# Kernel trace of the batch-1 decode step on the GPU box (run through gpurun):
#   bash tools/profile_decode.sh <tag> [decode_bench flags]  -> gpurun_out/prof_decode_<tag>/trace, then the per-kernel table and the
#   timeline of ONE replayed token (kernels, durations, gaps) in gpurun_out/<tag>_decode_kernels.txt / <tag>_decode_token_trace.txt
set -u
TAG=${1:-r04}
shift || true
OUT=gpurun_out/prof_decode_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/decode_bench.py --steps 32 "$@" > $OUT/bench.log 2>&1
tail -1 $OUT/bench.log | cut -c1-600
python3 tools/decode_trace_summary.py $OUT/trace gpurun_out/${TAG}_decode
