#!/usr/bin/env python3
"""The Perceiver connector's GEMMs (6 layers: 144 latents x 8 images = 1152 rows) under the lab library's tile modes: few rows against long K is a
launch of ~100 tiles on 256 CUs.   python tools/perceiver_gemm_ab.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aki_amd import ops, _lib
lib = _lib.load_lab(); _lib._lib = lib
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s, sc=1.0: (torch.randn(*s, device=dev, generator=g) * sc).to(torch.bfloat16)
def loop_us(fn, iters=30):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); torch.cuda.synchronize(); a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3
modes = {0: "product", 1: "256x256", 2: "128x128", 3: "128x96", 5: "64x64 ring4"}
for name, M, N, K, act in (("ff2", 1152, 1152, 4608, 0), ("ff1 gelu", 1152, 4608, 1152, ops.ACT_GELU_ERF), ("q", 1152, 512, 1152, 0), ("kv", 5760, 1024, 1152, 0), ("out", 1152, 1152, 512, 0),
                           ("siglip out", 4608, 1152, 1152, 0), ("siglip fc2", 4608, 1152, 4352, 0)):
    x, w = rnd(M, K), rnd(N, K, sc=0.03)
    fn = lambda: ops.linear(x, w, act=act)
    ref, res = None, {}
    for m in modes:
        lib.aki_lab_set_gemm_tile(m)
        y = fn(); torch.cuda.synchronize()
        if ref is None: ref = y.clone()
        else: assert torch.equal(ref, y), (name, m)
        res[m] = []
    for _ in range(5):
        for m in modes:
            lib.aki_lab_set_gemm_tile(m); res[m].append(loop_us(fn))
    lib.aki_lab_set_gemm_tile(0)
    print(f"{name:10s} M{M} N{N} K{K}: " + "  ".join(f"{modes[m]} {sorted(v)[2]:6.1f}" for m, v in res.items()))
