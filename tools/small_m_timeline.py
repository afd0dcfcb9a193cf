#!/usr/bin/env python3
"""Timeline of ONE small-M GEMM launch (lab library, probe_block = -2: every workgroup stamps the chip-wide 100 MHz clock at its start, K-loop
begin, K-loop end and exit): dispatch ramp, K-loop length, epilogue / fold tails, and how the launch's wall time divides among them.
    python tools/small_m_timeline.py [--configs -1:1,0:3,3:3] [--M 655]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from aki_amd import _lib, ops

dev = "cuda"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--M", type=int, default=655)
    ap.add_argument("--configs", default="-1:1,0:3,3:3")
    a = ap.parse_args()
    lib = _lib.load_lab()
    _lib._lib = lib
    ops.SPLITK_WS_MIN_BYTES = 512 << 20
    g = torch.Generator(device=dev).manual_seed(0)
    rnd = lambda *s, sc=1.0: (torch.randn(*s, device=dev, generator=g) * sc).to(torch.bfloat16)
    M, d, F, NB = a.M, 3072, 8192, 12
    x, r, act = [rnd(M, d) for _ in range(NB)], [rnd(M, d) for _ in range(NB)], [rnd(M, F) for _ in range(NB)]
    wo, wd = [rnd(d, d, sc=0.02) for _ in range(NB)], [rnd(d, F, sc=0.02) for _ in range(NB)]
    y = torch.empty(M, d, device=dev, dtype=torch.bfloat16)
    st = ops.new_stats(M, dev)
    stamps = torch.zeros(4 * 8192, dtype=torch.int64, device=dev)
    cases = {"o_proj +res+stats": lambda i: ops.linear(x[i], wo[i], residual=r[i], stats_out=st, stats_eps=1e-5, out=y),
             "o_proj +res": lambda i: ops.linear(x[i], wo[i], residual=r[i], out=y),
             "o_proj plain": lambda i: ops.linear(x[i], wo[i], out=y),
             "down +res+stats": lambda i: ops.linear(act[i], wd[i], residual=r[i], stats_out=st, stats_eps=1e-5, out=y)}
    for name, fn in cases.items():
        for cfg in a.configs.split(","):
            v, ks = [int(t_) for t_ in cfg.split(":")]
            lib.aki_lab_set_small_m(v, ks)
            lib.aki_lab_set_clock_probe(None)
            for i in range(NB):
                fn(i)
            torch.cuda.synchronize()
            lib.aki_lab_set_clock_probe(stamps.data_ptr())
            lib.aki_lab_set_probe_block(-2)
            rows = []
            for rep in range(6):
                lib.aki_lab_set_clock_probe(None)
                fn((rep + 3) % NB)            # an unstamped predecessor of the same kind right before: the steady state of a layer loop
                lib.aki_lab_set_clock_probe(stamps.data_ptr())
                stamps.zero_()
                fn(rep % NB)
                torch.cuda.synchronize()
                s_ = stamps.view(-1, 4).cpu().numpy().astype(np.int64)
                s_ = s_[s_[:, 0] > 0]
                rows.append(s_)
            s_ = rows[-1]
            t0 = s_[:, 0].min()
            start, lb, le, end = [(s_[:, k] - t0) / 100.0 for k in range(4)]      # us
            n = len(s_)
            q = lambda v_, p_: float(np.percentile(v_, p_))
            print(f"{name:18s} v{v} k{ks}: {n} wgs | start p50 {q(start,50):5.1f} p90 {q(start,90):5.1f} max {start.max():5.1f} | prologue {q(lb-start,50):4.1f} | "
                  f"K loop p50 {q(le-lb,50):5.1f} max {float((le-lb).max()):5.1f} | tail p50 {q(end-le,50):4.1f} p90 {q(end-le,90):4.1f} max {float((end-le).max()):5.1f} | "
                  f"last exit {end.max():5.1f} us", flush=True)
    lib.aki_lab_set_small_m(-1, 1)
    lib.aki_lab_set_clock_probe(None)
    lib.aki_lab_set_probe_block(0)


if __name__ == "__main__":
    main()
