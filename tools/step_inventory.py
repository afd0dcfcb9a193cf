#!/usr/bin/env python3
"""Kernel inventory of ONE step of a rocprofv3 --kernel-trace CSV (a step = from one im2col launch to the next): launches, total and average time per kernel, wall and busy time.
    python tools/step_inventory.py gpurun_out/prof/x_kernel_trace.csv [--top 40]"""
import argparse, collections, csv, re
ap = argparse.ArgumentParser(); ap.add_argument("csv"); ap.add_argument("--top", type=int, default=40); a = ap.parse_args()
rows = sorted(csv.DictReader(open(a.csv)), key=lambda r: int(r["Start_Timestamp"]))
st = [i for i, r in enumerate(rows) if "im2col" in r["Kernel_Name"]]
lo, hi = st[-2], st[-1]
wall = (int(rows[hi]["Start_Timestamp"]) - int(rows[lo]["Start_Timestamp"])) / 1e6
agg = collections.defaultdict(lambda: [0, 0]); busy = 0
for r in rows[lo:hi]:
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); busy += d
    n = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")[:110]
    agg[n][0] += 1; agg[n][1] += d
print(f"step: wall {wall:.3f} ms, busy {busy / 1e6:.3f} ms, {hi - lo} launches")
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:a.top]:
    print(f"{c:5d} x {t / c / 1e3:8.1f} us = {t / 1e3:9.1f} us  {n}")
