#!/usr/bin/env python3
"""Does the row pitch of a cold weight matrix matter?  gate_up + SwiGLU and down at the benchmark shape, weights rotated through 12 buffers (from HBM),
the weight stored with row pitch K (as the model has it), K + 64, K + 128 and the next power of two.   python tools/weight_stride_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aki_amd import ops
dev = "cuda"; NB = 12
M = 8 * 655
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s, sc=1.0: (torch.randn(*s, device=dev, generator=g) * sc).to(torch.bfloat16)
def bench(name, N, K, act, n_out, pitches):
    x = rnd(M, K); y = torch.empty(M, n_out, device=dev, dtype=torch.bfloat16)
    res = {}
    bufs = {p: [rnd(N, p, sc=0.02) for _ in range(NB)] for p in pitches}
    def run(p, rot, iters=2 * NB):
        evs = []
        for i in range(iters):
            w = bufs[p][i % NB if rot else 0][:, :K]
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); ops.linear(x, w, act=act, out=y); b.record(); evs.append((a, b))
        torch.cuda.synchronize()
        return sum(a.elapsed_time(b) for a, b in evs) / iters * 1e3
    for p in pitches:
        run(p, 1, NB)
    out = {}
    for p in pitches:
        hot = min(run(p, 0) for _ in range(3)); cold = min(run(p, 1) for _ in range(3))
        out[p] = (round(hot, 1), round(cold, 1))
    print(name, {f"pitch {p}": v for p, v in out.items()}, "(hot us, cold us)")
bench("gate_up + SwiGLU", 16384, 3072, ops.ACT_SWIGLU, 8192, (3072, 3136, 3200, 4096))
bench("down", 3072, 8192, 0, 3072, (8192, 8256, 8320))
bench("o_proj", 3072, 3072, 0, 3072, (3072, 3136, 3200, 4096))
