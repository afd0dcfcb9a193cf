#!/usr/bin/env python3
"""Multi-GPU day-one table: from the JSON lines of `bench.py --gpus N` (and optionally `tools/train_bench.py --gpus N`) at N = 1, 2, 4, 8
produce what BASELINE.json's north_star asks for -

  forward:   image+text tokens/s at each N, x-scaling against N = 1, parallel efficiency, per-rank min / max of ms per step (a straggler
             shows there), and the fraction of the attention roofline the MMA op reached at each N;
  training:  tokens/s and x-scaling of the pre-training step (the reference's DDP step: train/train.py:311-312, train/distributed.py:99-111),
             the gradient exchange alone / exposed behind the backward pass / overlap fraction, per-bucket timeline (when a bucket's last
             gradient was written, when its sum was in place), and the exposed exchange against the two projections of DESIGN.md section 6:
             one-hop direct reduce-scatter + all-gather over the 7 xGMI links (13 ms for 7.8 GB of bf16 gradients) and a single ring (45 ms).

    python tools/scale_report.py SCALE_r06.json [train_lines.jsonl] [--md]

Input: a file with one JSON object per line, or one JSON document that contains such objects anywhere (the driver's SCALE_rNN.json wraps
them: every dict with "n_gpus" and "value" is taken).  No GPU needed.  The 13 / 45 ms projections scale with the exchanged bytes."""
import json
import sys

XGMI_LINKS, XGMI_GBS = 7, 153.0          # per GPU, point to point (MI355X guide)
DIRECT_MS_PER_GB = 13.0 / 7.8            # reduce-scatter + all-gather, every peer one hop away, all links busy
RING_MS_PER_GB = 45.0 / 7.8              # one ring: per-link bound


def walk(o):
    if isinstance(o, dict):
        if "n_gpus" in o and "value" in o:
            yield o
        for v in o.values():
            yield from walk(v)
    elif isinstance(o, list):
        for v in o:
            yield from walk(v)


def load(path):
    txt = open(path).read().strip()
    objs = []
    try:
        objs = list(walk(json.loads(txt)))
    except json.JSONDecodeError:
        for line in txt.split("\n"):
            line = line.strip()
            if line.startswith("{"):
                try:
                    objs += list(walk(json.loads(line)))
                except json.JSONDecodeError:
                    pass
    return objs


def forward_table(rows):
    rows = sorted((r for r in rows if "training" not in r.get("metric", "")), key=lambda r: r["n_gpus"])
    if not rows:
        return []
    base = next((r for r in rows if r["n_gpus"] == 1), rows[0])
    out = []
    for r in rows:
        n = r["n_gpus"]
        x = r["value"] / base["value"] * base["n_gpus"]
        mk = r.get("mma_kernel") or {}
        out.append({"n_gpus": n, "rccl_ranks": r.get("rccl_ranks"), "tokens_per_s": r["value"], "ms_per_step": r.get("ms_per_step"),
                    "x_scaling": round(x, 3), "efficiency": round(x / n, 3),
                    "rank_min_ms": r.get("ms_per_step_rank_min"), "rank_max_ms": r.get("ms_per_step_rank_max"),
                    "attention_roofline_frac": mk.get("frac"), "dominant_kernel_frac": (r.get("roofline") or {}).get("frac")})
    return out


def train_table(rows):
    rows = sorted((r for r in rows if "training" in r.get("metric", "")), key=lambda r: r["n_gpus"])
    if not rows:
        return []
    base = next((r for r in rows if r["n_gpus"] == 1), rows[0])
    out = []
    for r in rows:
        n = r["n_gpus"]
        x = r["value"] / base["value"] * base["n_gpus"]
        gb = (r.get("exchange_bytes") or 0) / 1e9
        row = {"n_gpus": n, "rccl_ranks": r.get("rccl_ranks"), "tokens_per_s": r["value"], "ms_per_step": r.get("ms_per_step"), "x_scaling": round(x, 3),
               "efficiency": round(x / n, 3), "exchange_GB": round(gb, 2), "exchange_alone_ms": r.get("exchange_ms"),
               "exchange_exposed_ms": r.get("exchange_exposed_ms"), "overlap_frac": r.get("overlap_frac"),
               "projection_direct_ms": round(gb * DIRECT_MS_PER_GB, 1) if n > 1 else 0.0, "projection_ring_ms": round(gb * RING_MS_PER_GB, 1) if n > 1 else 0.0,
               "parts_ms": r.get("parts_ms")}
        if r.get("exchange_ms") and n > 1:
            row["exchange_alone_vs_direct"] = round(r["exchange_ms"] / max(gb * DIRECT_MS_PER_GB, 1e-9), 2)
            row["exchange_alone_vs_ring"] = round(r["exchange_ms"] / max(gb * RING_MS_PER_GB, 1e-9), 2)
        tl = r.get("bucket_timeline")
        if tl and tl.get("buckets"):
            end = tl.get("backward_compute_end_ms")
            bs = tl["buckets"]
            row["buckets"] = [{"index": b["index"], "MiB": b["MiB"], "launched_ms": b["launched_ms"], "complete_ms": b["complete_ms"],
                               "in_flight_ms": round(b["complete_ms"] - b["launched_ms"], 3),
                               "exposed_ms": round(max(0.0, b["complete_ms"] - end), 3) if end is not None else None} for b in bs]
            row["backward_compute_end_ms"] = end
            row["last_bucket_complete_ms"] = max(b["complete_ms"] for b in bs)
        out.append(row)
    return out


def md(fw, tr):
    lines = []
    if fw:
        lines += ["| GPUs | tokens/s | x-scaling | efficiency | ms/step | rank min / max ms | MMA op, fraction of MFMA peak |", "|---|---|---|---|---|---|---|"]
        for r in fw:
            lines.append(f"| {r['n_gpus']} | {r['tokens_per_s']:.0f} | {r['x_scaling']:.2f} | {r['efficiency']:.2f} | {r['ms_per_step']} | "
                         f"{r['rank_min_ms']} / {r['rank_max_ms']} | {r['attention_roofline_frac']} |")
    if tr:
        lines += ["", "| GPUs | train tokens/s | x-scaling | ms/step | exchange GB | alone ms | exposed ms | overlap | direct / ring projection ms |", "|---|---|---|---|---|---|---|---|---|"]
        for r in tr:
            lines.append(f"| {r['n_gpus']} | {r['tokens_per_s']:.0f} | {r['x_scaling']:.2f} | {r['ms_per_step']} | {r['exchange_GB']} | {r['exchange_alone_ms']} | "
                         f"{r['exchange_exposed_ms']} | {r['overlap_frac']} | {r['projection_direct_ms']} / {r['projection_ring_ms']} |")
    return "\n".join(lines)


def main(argv):
    paths = [a for a in argv if not a.startswith("--")]
    if not paths:
        print(__doc__)
        return 2
    rows = []
    for p in paths:
        rows += load(p)
    fw, tr = forward_table(rows), train_table(rows)
    if "--md" in argv:
        print(md(fw, tr))
    else:
        print(json.dumps({"forward": fw, "training": tr, "target": {"forward_x_at_8": 3.5},
                          "met": (None if not any(r["n_gpus"] == 8 for r in fw) else bool(next(r for r in fw if r["n_gpus"] == 8)["x_scaling"] >= 3.5))}, indent=1))
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
