#!/usr/bin/env python3
"""Where a cold weight matrix costs its time: the hand-placed lab loop's phase stamps (tile mode 4, workgroup 0) and the 8-wave kernel's prologue /
K loop / epilogue cycles, with the weight re-used every launch (on-die) or rotated through 12 buffers (from HBM).   python tools/gemm_cold_stamps.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aki_amd import _lib, ops
dev = "cuda"; NB = 12
lib = _lib.load_lab(); _lib._lib = lib
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s, sc=1.0: (torch.randn(*s, device=dev, generator=g) * sc).to(torch.bfloat16)
probe = torch.zeros(32, dtype=torch.int64, device=dev)
M = 8 * 655
for name, N, K in (("gate_up-like plain", 16384, 3072), ("down", 3072, 8192)):
    x = rnd(M, K); ws = [rnd(N, K, sc=0.02) for _ in range(NB)]
    y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    nk = K // 64
    for mode in (4, 1):
        for cold in (0, 1):
            lib.aki_lab_set_gemm_tile(mode)
            lib.aki_lab_set_clock_probe(probe.data_ptr())
            acc = torch.zeros(32, dtype=torch.float64)
            times = []
            for i in range(3 * NB):
                w = ws[i % NB] if cold else ws[0]
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); ops.linear(x, w, out=y); b.record()
                torch.cuda.synchronize()
                if i >= NB:
                    acc += probe.to(torch.float64).cpu(); times.append(a.elapsed_time(b) * 1e3)
            v = (acc / (2 * NB)).tolist()
            t = sorted(times)[len(times) // 2]
            if mode == 4:
                ph = [sum(v[2 + 4 * wv + i] for wv in range(4)) / 4 / nk for i in range(4)]
                print(f"{name:20s} lab loop  {'cold W' if cold else 'hot W ':6s}: {t:7.1f} us; per K-step: top wait+barrier {ph[0]:6.0f}  half-step 0 {ph[1]:6.0f}  mid wait+barrier {ph[2]:6.0f}  half-step 1 {ph[3]:6.0f}; "
                      f"prologue {v[18]:6.0f} epilogue {v[19]:6.0f} lifetime {v[0]:8.0f}")
            else:
                print(f"{name:20s} 8-wave    {'cold W' if cold else 'hot W ':6s}: {t:7.1f} us; K loop / step {(v[0] - v[18] - v[19]) / nk:6.0f}; prologue {v[18]:6.0f} epilogue {v[19]:6.0f} lifetime {v[0]:8.0f}")
lib.aki_lab_set_clock_probe(None); lib.aki_lab_set_gemm_tile(0)
