#!/bin/bash
# PMC passes over the attention core alone (tools/attn_only.py): where do the wave cycles go?   bash tools/attn_pmc.sh
# Counters in their own runs, no trace domains alongside (MI355X guide).  Prints per-grid averages of every counter.
set -u
OUT=gpurun_out/attn_pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_TRANS" "SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_WAVES" "GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_COEXEC_CYCLES"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -- python3 tools/attn_only.py ${CASE:-0} > $OUT/p$i.log 2>&1
  tail -1 $OUT/p$i.log | cut -c1-200
done
python3 - <<'PY'
import csv, glob, collections
tot = collections.defaultdict(dict)
for d in sorted(glob.glob('gpurun_out/attn_pmc/p*/')):
    for f in glob.glob(d + '**/*counter_collection.csv', recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(int)
        for r in csv.DictReader(open(f)):
            if 'mma_attn_bf16' not in r['Kernel_Name']: continue
            key = r['Grid_Size']
            acc[key][r['Counter_Name']] += float(r['Counter_Value']); n[(key, r['Counter_Name'])] += 1
        for key in acc:
            for c, v in acc[key].items():
                tot[key][c] = round(v / n[(key, c)])
for key in tot:
    print('grid', key, tot[key])
PY
rm -rf $OUT/p*/
