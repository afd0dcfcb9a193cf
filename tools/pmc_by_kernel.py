#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counters per kernel name (first 70 characters) and print them per launch.
    python tools/pmc_by_kernel.py <rocprofv3 output dir> [substring]"""
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
want = sys.argv[2] if len(sys.argv) > 2 else "gemm"
agg = collections.defaultdict(lambda: collections.defaultdict(float)); launches = collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"][:70]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); launches[k].add(r["Dispatch_Id"])
for k, d in agg.items():
    if want in k:
        n = len(launches[k])
        print(f"{k}  ({n} launches)  " + "  ".join(f"{c} {v / n:.4g}" for c, v in sorted(d.items())))
