#!/usr/bin/env python3
"""A/B of the batched-decode GEMMs with the RMSNorm as a launch of its own vs in the skinny GEMM's prologue (cold weights: 12 rotating sets)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from aki_amd import ops
dev = "cuda"
NB = 12
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s, sc=1.0: (torch.randn(*s, device=dev, generator=g) * sc).to(torch.bfloat16)


def timed(fn, reps=4):
    for i in range(NB):
        fn(i)
    torch.cuda.synchronize()
    best = 1e30
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for k in range(reps * NB):
            fn(k % NB)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / (reps * NB) * 1e3)
    return best


for M in (2, 4, 8):
    x = rnd(M, 3072)
    nw = rnd(3072, sc=0.1) + 1
    for name, N, act in (("qkv N9216", 9216, ops.ACT_NONE), ("gate_up N16384", 16384, ops.ACT_SWIGLU), ("lm_head N32064", 32064, ops.ACT_NONE)):
        ws = [rnd(N, 3072, sc=0.02) for _ in range(NB)]
        two = timed(lambda i: ops.linear(ops.rmsnorm(x, nw, 1e-5), ws[i], act=act))
        one = timed(lambda i: ops.decode_linear(x, ws[i], nw, 1e-5, act=act))
        bare = timed(lambda i: ops.linear(x, ws[i], act=act))
        print(f"M {M} {name:16s}: norm launch + GEMM {two:6.1f} us, norm in the prologue {one:6.1f} us, GEMM alone {bare:6.1f} us", flush=True)
        del ws
