#!/usr/bin/env python3
"""aki_gemm_tn against the row pitch of its operands: 64 rows x 512 B per K-step at a pitch of 18 KB (dqkv) or 32 KB (d gate_up) land on few memory channels.
    python tools/gemm_tn_pitch_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aki_amd import train_ops as T
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for name, M, N, K in (("qkv", 5240, 9216, 3072), ("o_proj", 5240, 3072, 3072), ("gate_up", 5240, 16384, 3072), ("down", 5240, 3072, 8192)):
    line = f"{name:8s} dW [{N} x {K}]:"
    for pad_a, pad_b in ((0, 0), (64, 0), (256, 0), (1024, 0), (0, 256), (256, 256)):
        dyb = torch.randn(M, N + pad_a, device=dev, generator=g).to(torch.bfloat16); xb = torch.randn(M, K + pad_b, device=dev, generator=g).to(torch.bfloat16)
        dy, x = dyb[:, :N], xb[:, :K]
        line += f"  dY+{pad_a}/X+{pad_b}: {timeit(lambda: T.gemm_tn(dy, x)):6.1f}"
    print(line + "  us")
