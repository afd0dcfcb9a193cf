"""Per-kernel table and the timeline of one replayed decode token out of a rocprofv3 --kernel-trace --stats tree.
usage: python tools/decode_trace_summary.py <trace dir> <output prefix>"""
import csv, glob, os, sys

src, dst = sys.argv[1], sys.argv[2]
short = lambda n: (n.replace("void ", "")[:118])
stats = glob.glob(os.path.join(src, "*", "*_kernel_stats.csv"))
with open(dst + "_kernels.txt", "w") as f:
    for r in (csv.DictReader(open(stats[0])) if stats else []):
        f.write(f"{short(r['Name']):120s} {int(r['Calls']):7d} {float(r['AverageNs']) / 1e3:10.2f} us {float(r['Percentage']):6.2f} %\n")
tr = glob.glob(os.path.join(src, "*", "*_kernel_trace.csv"))
rows = sorted(csv.DictReader(open(tr[0])), key=lambda r: int(r["Start_Timestamp"])) if tr else []
# one token = from the end of the second-last big step kernel (decode_chain or the last gemv of a step) to the end of the last one;
# anchor on the lm_head GEMV (the widest N), which closes every step in both paths
is_anchor = lambda r: "decode_chain_kernel" in r["Kernel_Name"]
idx = [i for i, r in enumerate(rows) if is_anchor(r)]
with open(dst + "_token_trace.txt", "w") as f:
    if len(idx) >= 3:
        a, b = idx[-3], idx[-2]
        t_prev = int(rows[a]["Start_Timestamp"])
        f.write(f"one greedy token: {(int(rows[b]['Start_Timestamp']) - t_prev) / 1e3:.1f} us from chain start to chain start\n")
        end_prev = None
        for r in rows[a:b]:
            s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            gap = "" if end_prev is None else f"gap {(s - end_prev) / 1e3:7.2f} us"
            f.write(f"{(s - t_prev) / 1e3:9.2f} us  {(e - s) / 1e3:9.2f} us  {gap:18s} {short(r['Kernel_Name'])[:100]}\n")
            end_prev = e
    else:
        f.write("no decode_chain_kernel dispatches in the trace\n")
print(open(dst + "_token_trace.txt").read())
