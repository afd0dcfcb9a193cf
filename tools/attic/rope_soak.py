#!/usr/bin/env python3
"""QKV + RoPE stage and the fused attention op, launch after launch on the same inputs: q / k / v and the attention output must be bit-identical with the first
launch, under the product's tile choice and under the forced tiles of the lab library.   python tools/rope_soak.py [--seconds 20]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aki_amd import ops, _lib
ap = argparse.ArgumentParser(); ap.add_argument("--seconds", type=float, default=20.0); a = ap.parse_args()
lib = _lib.load_lab(); _lib._lib = lib
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
B, L, H, Dh = 8, 655, 32, 96
x = torch.randn(B, L, H * Dh, device=dev, generator=g).to(torch.bfloat16); w = (torch.randn(3 * H * Dh, H * Dh, device=dev, generator=g) * 0.02).to(torch.bfloat16)
cos, sin = torch.randn(L, Dh, device=dev, generator=g), torch.randn(L, Dh, device=dev, generator=g)
rs = torch.rand(B * L, device=dev, generator=g) + 0.5
table = ops.MaskTable.causal(B, L, dev)
for mode, name in ((0, "product"), (1, "256x256"), (2, "128x128"), (2048, "two-stage loops")):
    lib.aki_lab_set_gemm_tile(mode)
    ref = None; n = bad = 0; t0 = time.time()
    while time.time() - t0 < a.seconds:
        outs = [(ops.qkv_rope(x, w, cos, sin, H, row_scale=rs), ops.mma_attn(x, w, cos, sin, table, H, row_scale=rs)) for _ in range(3)]
        for (q, k, v), o in outs:
            n += 1
            if ref is None: ref = (q.clone(), k.clone(), v.clone(), o.clone())
            elif not (torch.equal(q, ref[0]) and torch.equal(k, ref[1]) and torch.equal(v, ref[2]) and torch.equal(o, ref[3])): bad += 1
    print(f"{name:18s} {n:7d} launch pairs, {bad} differ from the first", flush=True)
lib.aki_lab_set_gemm_tile(0)
