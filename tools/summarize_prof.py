"""Summarise a rocprofv3 output tree (kernel-trace --stats + separate --pmc passes) into profiles/<tag>_*.
usage: python tools/summarize_prof.py gpurun_out/prof_r1 profiles/r01"""
import csv, glob, os, sys, collections, json

src, dst = sys.argv[1], sys.argv[2]
os.makedirs(os.path.dirname(dst) or ".", exist_ok=True)


def short(n):
    n = n.replace("void ", "")
    return n if len(n) < 110 else n[:107] + "..."


def newest(pattern):
    """One rocprofv3 run per directory is expected; a directory merged back from several gpurun calls holds several (one file per
    process id): take the newest and say so, never an average over builds."""
    found = sorted(glob.glob(pattern), key=os.path.getmtime)
    if len(found) > 1:
        print(f"summarize_prof: {len(found)} runs under {os.path.dirname(os.path.dirname(pattern))}: using the newest, {found[-1]}", file=sys.stderr)
    return found[-1:]


stats = newest(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))
rows = list(csv.DictReader(open(stats[0]))) if stats else []
with open(dst + "_kernel_stats.csv", "w") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for r in rows:
        w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])

pmc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in glob.glob(os.path.join(src, "pmc_*")):
    for fcsv in newest(os.path.join(d, "*", "*_counter_collection.csv")):
        for r in csv.DictReader(open(fcsv)):
            pmc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = []
for k, cs in pmc.items():
    if "aki::" not in k and "attn_fwd" not in k:
        continue
    avg = {c: sum(v) / len(v) for c, v in cs.items()}
    e = {"kernel": k, "launches": len(next(iter(cs.values())))}
    if "FETCH_SIZE" in avg:
        e["FETCH_SIZE_KiB_raw"] = round(avg["FETCH_SIZE"], 1)
        e["hbm_read_bytes_corrected_x2"] = int(avg["FETCH_SIZE"] * 1024 * 2)   # gfx950: FETCH_SIZE reports 1/2 of wide streams
    if "WRITE_SIZE" in avg:
        e["hbm_write_bytes"] = int(avg["WRITE_SIZE"] * 1024)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in avg and avg.get("GRBM_GUI_ACTIVE", 0) > 0:
        e["SQ_VALU_MFMA_BUSY_CYCLES"] = int(avg["SQ_VALU_MFMA_BUSY_CYCLES"])
        e["GRBM_GUI_ACTIVE"] = int(avg["GRBM_GUI_ACTIVE"])
        e["SQ_BUSY_CYCLES"] = int(avg.get("SQ_BUSY_CYCLES", 0))
        e["SQ_WAVE_CYCLES"] = int(avg.get("SQ_WAVE_CYCLES", 0))
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs; MFMA busy cycles over all 1024 SIMDs
        e["mfma_busy_frac_of_simd_cycles"] = round(avg["SQ_VALU_MFMA_BUSY_CYCLES"] / (avg["GRBM_GUI_ACTIVE"] / 8.0 * 256 * 4), 4)
    out.append(e)
out.sort(key=lambda e: e["kernel"])
# which tree these counters were measured on: bench.py prints "traffic_stale": true when the running tree's kernels differ
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aki_amd.build import csrc_hash
out.insert(0, {"kernel": "__meta__", "csrc_tree_hash": csrc_hash()})
json.dump(out, open(dst + "_pmc_summary.json", "w"), indent=1)
for e in out:
    print(json.dumps(e))
