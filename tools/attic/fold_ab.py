#!/usr/bin/env python3
"""A/B of the folded normalisation on the benchmark workload, same process and box: forward ms/step with the RMSNorm /
LayerNorm launches (fold off) and with the norms folded into the GEMMs (fold on), plus the HOST time to issue one forward
(no sync) - if that approaches the GPU time the step is launch-bound and kernel savings do not show.

    python tools/fold_ab.py [--steps 10]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from aki_amd.factory import build_aki  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=10)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    model = build_aki(dtype=torch.bfloat16, device=dev, seed=0).eval()
    vx, ids, am = bench.synth_batch(bench.BATCH, dev, torch.bfloat16, model.media_token_id, seed=1000)

    def set_fold(on):
        model.lang_model.model.fold_norms = on
        model.vision_encoder.encoder.fold_norms = on

    def run(n):
        host = 0.0
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            h0 = time.perf_counter()
            with torch.no_grad():
                model(vx, ids, attention_mask=am)
            host += time.perf_counter() - h0
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3, host / n * 1e3

    for rnd in range(3):
        for on in (False, True):
            set_fold(on)
            run(3)
            ms, host = run(a.steps)
            print(f"round {rnd} fold={'on ' if on else 'off'}  {ms:7.3f} ms/step   host issue {host:7.3f} ms/step", flush=True)


if __name__ == "__main__":
    main()
