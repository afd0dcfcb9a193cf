#!/usr/bin/env python3
"""Soak test of the row-statistics hand-off (GEMM epilogue partials -> ticket -> last-arriver reduction, gemm_bf16.hip): thousands
of producer launches over changing shapes, each checked against the statistics of the output it wrote.  A visibility race would
show up as a sporadic mismatch; the protocol must also leave its counters at zero for the next launch whatever ran before.

    python tools/stats_stress.py [--seconds 60]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from aki_amd import ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=60.0)
    a = ap.parse_args()
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    shapes = [(5240, 3072, 3072), (5240, 3072, 8192), (4608, 1152, 1152), (4608, 1152, 4352), (1380, 3072, 256), (300, 512, 256), (37, 1152, 640),
              (8 * 655, 1152, 3072), (2000, 9216, 512)]
    data = {}
    for M, N, K in shapes:
        data[(M, N, K)] = ((torch.randn(M, K, device=dev, generator=g)).to(torch.bfloat16), (torch.randn(N, K, device=dev, generator=g) * 0.05).to(torch.bfloat16),
                           (torch.randn(M, N, device=dev, generator=g) * 2).to(torch.bfloat16))
    t0, n, worst = time.time(), 0, 0.0
    while time.time() - t0 < a.seconds:
        for (M, N, K), (x, w, r) in data.items():
            for ln in (False, True):
                outs = []
                for rep in range(4):                                  # several launches in flight on the same workspace
                    st = ops.new_stats(M, dev, ln=ln)
                    y = ops.linear(x, w, residual=r, stats_out=st, stats_eps=1e-6)
                    outs.append((y, st))
                for y, st in outs:
                    yf = y.float()
                    if ln:
                        mu = yf.mean(-1)
                        want = torch.rsqrt(yf.var(-1, unbiased=False) + 1e-6)
                        e2 = float(((st.mean - mu).abs() / (mu.abs() + 1e-3)).max())
                    else:
                        want, e2 = torch.rsqrt(yf.pow(2).mean(-1) + 1e-6), 0.0
                    e1 = float(((st.rstd - want).abs() / want).max())
                    worst = max(worst, e1)
                    if not (e1 < 1e-4 and e2 < 1e-3):
                        raise SystemExit(f"MISMATCH after {n} launches: shape {(M, N, K)} ln={ln}: rstd rel err {e1:.3g}, mean rel err {e2:.3g}")
                    n += 1
    print(f"{n} producer launches checked in {time.time() - t0:.0f} s, worst relative rstd error {worst:.2e}: OK")


if __name__ == "__main__":
    main()
