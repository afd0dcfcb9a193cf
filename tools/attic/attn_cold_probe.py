#!/usr/bin/env python3
"""The two attention kernels with q / k / v re-used (on-die) or rotated through 12 buffers (from HBM: in the forward they were written by the GEMM right before).
    python tools/attn_cold_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from aki_amd import ops
dev = "cuda"; NB = 12
g = torch.Generator(device=dev).manual_seed(0)
def run(fn, n, cold, iters=2 * NB):
    evs = []
    for i in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(i % n if cold else 0); b.record(); evs.append((a, b))
    torch.cuda.synchronize()
    return sum(a.elapsed_time(b) for a, b in evs) / iters * 1e3
B, H, L = 8, 32, 655
qs = [tuple(torch.randn(B, H, L, 96, device=dev, generator=g).to(torch.bfloat16) for _ in range(3)) for _ in range(NB)]
table = ops.MaskTable.from_host([[(6, 150, 150, 638)]] * B, np.ones((B, L)), None, dev)
f = lambda i: ops.mma_attn_core(*qs[i], table, 96 ** -0.5, dead_rows=0)
run(f, NB, 1, NB)
print("decoder core B8 H32 L655: hot", round(min(run(f, NB, 0) for _ in range(4)), 1), "us; cold", round(min(run(f, NB, 1) for _ in range(4)), 1), "us")
B, H, L, D = 8, 16, 576, 72
qkvs = [torch.randn(B, L, 3, H, D, device=dev, generator=g).to(torch.bfloat16) for _ in range(NB)]
f2 = lambda i: ops.attention(qkvs[i][:, :, 0], qkvs[i][:, :, 1], qkvs[i][:, :, 2], D ** -0.5)
run(f2, NB, 1, NB)
print("SigLIP attention B8 H16 L576 D72: hot", round(min(run(f2, NB, 0) for _ in range(4)), 1), "us; cold", round(min(run(f2, NB, 1) for _ in range(4)), 1), "us")
