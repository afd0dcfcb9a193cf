#!/bin/bash
# Kernel trace + PMC passes of the attention core alone at BASELINE configs[3]'s shape (L = 4096, 4 images), batch 1 and 4 (through gpurun):
#   bash tools/profile_attn64.sh <tag>   -> gpurun_out/prof_<tag>_b{1,4}/..., then
#   python tools/summarize_prof.py gpurun_out/prof_<tag>_b1 profiles/<tag>_attn64_b1   (and _b4)
# PMC counters are collected in their own runs (no trace domains alongside), as the MI355X guide prescribes.
set -u
TAG=${1:-r06}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for B in 1 4; do
  OUT=gpurun_out/prof_${TAG}_b$B
  mkdir -p $OUT
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/attn_core_run.py $B 4096 4 30 > $OUT/trace.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 tools/attn_core_run.py $B 4096 4 6 > $OUT/pmc_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 tools/attn_core_run.py $B 4096 4 6 > $OUT/pmc_write.log 2>&1
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $OUT/pmc_mfma -- python3 tools/attn_core_run.py $B 4096 4 6 > $OUT/pmc_mfma.log 2>&1
  tail -1 $OUT/trace.log
done
