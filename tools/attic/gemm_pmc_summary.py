#!/usr/bin/env python3
"""Summarise the SQ counter passes of tools/gemm_pmc.sh: per GEMM kernel (template arguments, grid), the mean of each
counter per launch and its share of SQ_WAVE_CYCLES.

    python tools/gemm_pmc_summary.py gpurun_out/gemm_pmc_<tag> > profiles/<round>_gemm_sq_counters.txt
"""
import collections
import csv
import glob
import sys


def main():
    root = sys.argv[1]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"{root}/pass*/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "gemm_bf16" not in k:
                continue
            agg[(k[k.index("<"):k.index(">") + 1], r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("SQ counters per launch (tools/gemm_only.py: the decoder's GEMMs at M = 5240).  SQ_WAVE_CYCLES, SQ_WAIT_*, SQ_ACTIVE_INST_* count")
    print("quad-cycles summed over waves; SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over SIMDs (MI355X_MICROARCH.md).")
    for key, c in agg.items():
        m = {k: sum(v) / len(v) for k, v in c.items()}
        wc = m.get("SQ_WAVE_CYCLES", 1.0)
        print(f"\ngemm_bf16_kernel{key[0]}  grid {key[1]} threads")
        for k, v in sorted(m.items()):
            print(f"   {k:32s} {v:14.0f}   {v / wc:6.3f} x WAVE_CYCLES   ({len(c[k])} launches)")


if __name__ == "__main__":
    main()
