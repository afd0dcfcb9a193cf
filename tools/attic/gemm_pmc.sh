#!/bin/bash
# SQ counter passes over tools/gemm_only.py (run through gpurun): where the big GEMM's wave cycles go.
#   bash tools/gemm_pmc.sh <tag>  -> gpurun_out/gemm_pmc_<tag>/pass{1,2,3}
set -u
TAG=${1:-a}
OUT=gpurun_out/gemm_pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT/pass1 -- python3 tools/gemm_only.py 4 > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_MFMA_MOPS_BF16 --output-format csv -d $OUT/pass2 -- python3 tools/gemm_only.py 4 > $OUT/p2.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/pass3 -- python3 tools/gemm_only.py 4 > $OUT/p3.log 2>&1
tail -2 $OUT/p1.log $OUT/p2.log $OUT/p3.log
ls $OUT/pass1/* | head
