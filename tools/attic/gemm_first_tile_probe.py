#!/usr/bin/env python3
"""How long a workgroup of a multi-round GEMM launch waits for its first tiles (lab library: the workgroup chosen with aki_lab_set_probe_block stamps
entry, K-loop begin, first tile landed, K-loop end), operands re-used (hot) or rotated through 12 buffers (cold).   python tools/gemm_first_tile_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aki_amd import ops, _lib
lib = _lib.load_lab(); _lib._lib = lib
dev = "cuda"; NB = 12
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s, sc=1.0: (torch.randn(*s, device=dev, generator=g) * sc).to(torch.bfloat16)
M = 8 * 655
probe = torch.zeros(32, dtype=torch.int64, device=dev)
lib.aki_lab_set_clock_probe(probe.data_ptr())
for name, N, K, act in (("gate_up swiglu", 16384, 3072, ops.ACT_SWIGLU), ("lm_head", 32064, 3072, 0)):
    xs = [rnd(M, K) for _ in range(NB)]; ws = [rnd(N, K, sc=0.02) for _ in range(NB)]
    y = torch.empty(M, N // 2 if act else N, device=dev, dtype=torch.bfloat16)
    for blk in (0, 300, 700, 1100):
        lib.aki_lab_set_probe_block(blk)
        for cold in (0, 1):
            acc = torch.zeros(32, dtype=torch.float64)
            for i in range(3 * NB):
                ops.linear(xs[i % NB if cold else 0], ws[i % NB if cold else 0], act=act, out=y); torch.cuda.synchronize()
                if i >= NB: acc += probe.to(torch.float64).cpu()
            v = (acc / (2 * NB)).tolist(); nk = K // 64
            print(f"{name:14s} workgroup {blk:5d} {'cold' if cold else 'hot '}: lifetime {v[0]:8.0f} prologue {v[18]:6.0f} first tile landed after {v[21]:6.0f}; K loop / step {(v[0]-v[18]-v[19])/nk:6.0f} epilogue {v[19]:6.0f}")
lib.aki_lab_set_probe_block(0); lib.aki_lab_set_clock_probe(None)
