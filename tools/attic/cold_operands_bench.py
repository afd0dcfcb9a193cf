#!/usr/bin/env python3
"""Which operand of a decoder GEMM is expensive when it is not on-die?  gate_up + SwiGLU and down at the benchmark shape with
each operand either re-used every iteration (hot: L2 / Infinity-Cache resident) or rotated through 12 buffers (cold: from HBM),
plus the case the real forward has for the activation: written by another kernel immediately before the launch.
Interleaved arms, per-launch HIP events.    python tools/cold_operands_bench.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from aki_amd import ops  # noqa: E402

dev = "cuda"
NB = 12
M, d, F = 8 * 655, 3072, 8192
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s, sc=1.0: (torch.randn(*s, device=dev, generator=g) * sc).to(torch.bfloat16)


def bench(name, N, K, act, n_out):
    ws = [rnd(N, K, sc=0.02) for _ in range(NB)]
    xs = [rnd(M, K) for _ in range(NB)]
    ys = [torch.empty(M, n_out, device=dev, dtype=torch.bfloat16) for _ in range(NB)]
    src = rnd(M, K)

    def run(rot_w, rot_x, rot_y, fresh_x, iters=2 * NB):
        tot = 0.0
        evs = []
        for i in range(iters):
            w = ws[i % NB] if rot_w else ws[0]
            x = xs[i % NB] if rot_x else xs[0]
            y = ys[i % NB] if rot_y else ys[0]
            if fresh_x:
                x.copy_(src)                      # the activation is produced right before the launch, as in the forward
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            ops.linear(x, w, act=act, out=y)
            b.record()
            evs.append((a, b))
        torch.cuda.synchronize()
        return sum(a.elapsed_time(b) for a, b in evs) / iters * 1e3

    arms = {"all hot": (0, 0, 0, 0), "W cold": (1, 0, 0, 0), "X cold": (0, 1, 0, 0), "Y cold": (0, 0, 1, 0), "X fresh (hot buffer)": (0, 0, 0, 1),
            "X fresh (rotating)": (0, 1, 0, 1), "W cold + X fresh rot + Y cold": (1, 1, 1, 1), "W + X + Y cold": (1, 1, 1, 0)}
    res = {k: [] for k in arms}
    for _ in range(2):
        for k, v in arms.items():
            run(*v, iters=NB)
    for r in range(4):
        for k, v in arms.items():
            res[k].append(run(*v))
    base = min(res["all hot"])
    print(f"{name}  (M {M}, N {N}, K {K})")
    for k, v in res.items():
        print(f"   {k:32s} {min(v):7.1f} us  (+{min(v) - base:5.1f})", flush=True)


bench("gate_up + SwiGLU", 2 * F, d, ops.ACT_SWIGLU, F)
bench("down", d, F, 0, d)
bench("o_proj", d, d, 0, d)
