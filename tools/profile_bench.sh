#!/bin/bash
# Kernel trace + PMC passes of the headline benchmark on the GPU box (run through gpurun):
#   bash tools/profile_bench.sh <tag>      -> gpurun_out/prof_<tag>/{trace,pmc_fetch,pmc_write,pmc_mfma}, then
#   python tools/summarize_prof.py gpurun_out/prof_<tag> profiles/<tag>
# PMC counters are collected in their own runs (no trace domains alongside), as the MI355X guide prescribes.
set -u
TAG=${1:-r01}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > $OUT/bench_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $OUT/pmc_mfma -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > $OUT/pmc_mfma.log 2>&1
tail -1 $OUT/bench_trace.log | cut -c1-300
