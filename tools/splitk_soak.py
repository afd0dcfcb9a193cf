#!/usr/bin/env python3
"""Soak of the split-K fold (gemm_bf16.hip, template SK): the o_proj / down launches of a one-sample prefill (M = 655 and 207, residual + row
statistics) thousands of times on rotating cold operands, every output and every statistics vector compared bit for bit with the first launch on
the same operands.  The fold adds the slices' partial sums in slice order whatever the arrival order: one differing bit is a race.
    python tools/splitk_soak.py [--launches 4000]"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aki_amd import ops

dev = "cuda"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--launches", type=int, default=4000)
    a = ap.parse_args()
    g = torch.Generator(device=dev).manual_seed(0)
    rnd = lambda *s, sc=1.0: (torch.randn(*s, device=dev, generator=g) * sc).to(torch.bfloat16)
    d, F, NB = 3072, 8192, 6
    res = []
    for M in (655, 207):
        for name, K in (("o_proj", d), ("down", F)):
            x, r, w = [rnd(M, K) for _ in range(NB)], [rnd(M, d) for _ in range(NB)], [rnd(d, K, sc=0.02) for _ in range(NB)]
            ref = []
            for i in range(NB):
                st = ops.new_stats(M, dev)
                y = ops.linear(x[i], w[i], residual=r[i], stats_out=st, stats_eps=1e-5)
                ref.append((y.clone(), st.rstd.clone()))
            bad = torch.zeros((), dtype=torch.int64, device=dev)
            for n in range(a.launches):
                i = n % NB
                st = ops.new_stats(M, dev)
                y = ops.linear(x[i], w[i], residual=r[i], stats_out=st, stats_eps=1e-5)
                bad += (y != ref[i][0]).sum() + (st.rstd != ref[i][1]).sum()
            torch.cuda.synchronize()
            res.append({"M": M, "gemm": name, "K": K, "launches": a.launches, "differing_elements": int(bad)})
            print(json.dumps(res[-1]), flush=True)
    assert all(r_["differing_elements"] == 0 for r_ in res)


if __name__ == "__main__":
    main()
