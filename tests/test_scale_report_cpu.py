"""tools/scale_report.py: the multi-GPU day-one table (north_star: tokens/s at 1, 2, 4, 8 GPUs, x-scaling, per-rank spread, fraction of the
attention roofline; for the training step the exposed gradient exchange per bucket against DESIGN.md section 6's projections) from the JSON
lines bench.py / tools/train_bench.py print.  Synthetic lines here; tests/test_dp_gpu.py feeds it the real two-rank lines."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _fw(n, value, lo, hi):
    return {"metric": "image+text tokens/sec forward, AKI-4B, 336px img + 512 txt", "value": value, "unit": "tokens/s", "n_gpus": n, "rccl_ranks": n, "steps": 10,
            "ms_per_step": hi, "ms_per_step_rank_min": lo, "ms_per_step_rank_max": hi, "roofline": {"frac": 0.54}, "mma_kernel": {"frac": 0.42}}


def _tr(n, value, exch, exposed):
    return {"metric": "training tokens/s, AKI-4B pre-training step (fwd+bwd+all-reduce+clip+AdamW)", "value": value, "n_gpus": n, "rccl_ranks": n, "ms_per_step": 5240 * n / value * 1e3,
            "exchange_ms": exch, "exchange_exposed_ms": exposed, "overlap_frac": None if not exch else round(1 - exposed / exch, 3), "exchange_bytes": 7.8e9,
            "parts_ms": {"forward": 44.0, "backward": 87.0, "optimizer": 23.0},
            "bucket_timeline": {"backward_compute_end_ms": 87.0, "buckets": [{"index": 0, "MiB": 64.0, "launched_ms": 86.5, "complete_ms": 89.0},
                                                                               {"index": 1, "MiB": 512.0, "launched_ms": 70.0, "complete_ms": 80.0}]}}


def test_scale_report_tables(tmp_path):
    rows = [_fw(1, 130000.0, 40.3, 40.3), _fw(2, 255000.0, 40.5, 41.1), _fw(4, 500000.0, 40.4, 41.9), _fw(8, 980000.0, 40.6, 42.8),
            _tr(1, 33900.0, 0.0, 0.0), _tr(8, 250000.0, 16.0, 2.5)]
    scale = tmp_path / "SCALE.json"
    scale.write_text(json.dumps({"runs": [{"n": r["n_gpus"], "parsed": r} for r in rows[:4]]}))       # wrapped, the way a driver record nests the lines
    train = tmp_path / "train.jsonl"
    train.write_text("\n".join(json.dumps(r) for r in rows[4:]) + "\n")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "scale_report.py"), str(scale), str(train)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr
    rep = json.loads(r.stdout)
    fw = {row["n_gpus"]: row for row in rep["forward"]}
    assert fw[1]["x_scaling"] == 1.0 and abs(fw[8]["x_scaling"] - 980000 / 130000) < 1e-3 and abs(fw[8]["efficiency"] - 980000 / 130000 / 8) < 1e-3
    assert fw[4]["rank_min_ms"] == 40.4 and fw[4]["rank_max_ms"] == 41.9 and fw[2]["attention_roofline_frac"] == 0.42
    assert rep["met"] is True                                                             # >= 3.5x at 8 GPUs
    tr = {row["n_gpus"]: row for row in rep["training"]}
    assert abs(tr[8]["projection_direct_ms"] - 13.0) < 0.05 and abs(tr[8]["projection_ring_ms"] - 45.0) < 0.05 and tr[1]["projection_direct_ms"] == 0.0
    assert abs(tr[8]["exchange_alone_vs_direct"] - 16.0 / 13.0) < 0.02
    b = {x["index"]: x for x in tr[8]["buckets"]}
    assert b[0]["exposed_ms"] == 2.0 and b[1]["exposed_ms"] == 0.0 and b[1]["in_flight_ms"] == 10.0      # the small first bucket is the exposed tail
    md = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "scale_report.py"), str(scale), str(train), "--md"], capture_output=True, text=True, timeout=60)
    assert md.returncode == 0 and "| 8 | 980000 |" in md.stdout and "13.0 / 45.0" in md.stdout
