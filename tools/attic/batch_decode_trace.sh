#!/bin/bash
# Kernel stats + the timeline of ONE replayed decode step at batch B (the five-launch-per-layer path under graph replay):
#   bash tools/attic/batch_decode_trace.sh 8
set -u
B=${1:-8}
OUT=gpurun_out/prof_decode_b$B
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/decode_bench.py --batch $B --steps 24 > $OUT/bench.log 2>&1
tail -1 $OUT/bench.log | cut -c1-600
python3 - <<PY
import csv, glob
tr = glob.glob("$OUT/trace/*/*_kernel_trace.csv")
rows = sorted(csv.DictReader(open(tr[0])), key=lambda r: int(r["Start_Timestamp"]))
# a step ends with the widest GEMM (lm_head): anchor on the last three occurrences of the kernel with the largest grid among the skinny / gemv ones
heads = [i for i, r in enumerate(rows) if ("skinny_gemm" in r["Kernel_Name"] or "gemv_bf16" in r["Kernel_Name"]) and int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1) > 1500]
a, b = heads[-3], heads[-2]
step = rows[a + 1:b + 1]
t0 = int(rows[a]["End_Timestamp"])
agg = {}
busy = 0
for r in step:
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    busy += d
    k = r["Kernel_Name"].replace("void ", "")[:90]
    agg.setdefault(k, [0, 0])
    agg[k][0] += 1
    agg[k][1] += d
win = int(step[-1]["End_Timestamp"]) - t0
with open("gpurun_out/decode_b${B}_step.txt", "w") as f:
    f.write(f"one replayed decode step at batch $B: {len(step)} launches, window {win / 1e3:.1f} us, busy {busy / 1e3:.1f} us, idle {(win - busy) / 1e3:.1f} us\n")
    for k, (n, d) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        f.write(f"  {n:4d} x {d / n / 1e3:8.2f} us = {d / 1e3:9.1f} us  {k}\n")
print(open("gpurun_out/decode_b${B}_step.txt").read())
PY
