#!/usr/bin/env python3
"""The decode attention kernel launch after launch on the same cache: the output must be bit-identical with the first launch (its cross-wave merge reads
paired 64-bit values back from LDS - the instruction form that lost a half in the GEMM's statistics fold).   python tools/decode_attn_soak.py [--seconds 20]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aki_amd import ops
ap = argparse.ArgumentParser(); ap.add_argument("--seconds", type=float, default=20.0); a = ap.parse_args()
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
for B, H, Dh, cap, nk in ((1, 32, 96, 1024, 655), (8, 32, 96, 1024, 700), (16, 32, 96, 4096, 3000)):
    q = torch.randn(B, H, Dh, device=dev, generator=g).to(torch.bfloat16)
    kc = torch.randn(B, H, cap, Dh, device=dev, generator=g).to(torch.bfloat16); vc = torch.randn(B, H, cap, Dh, device=dev, generator=g).to(torch.bfloat16)
    n_keys = torch.full((B,), nk, dtype=torch.int32, device=dev)
    ws = ops.decode_attn_workspace(B, H, Dh, cap, dev)
    ref = None; n = bad = 0; t0 = time.time()
    while time.time() - t0 < a.seconds / 3:
        outs = [ops.decode_attn(q, kc, vc, n_keys, Dh ** -0.5, max_keys=nk, ws=ws) for _ in range(8)]
        for o in outs:
            n += 1
            if ref is None: ref = o.clone()
            elif not torch.equal(o, ref): bad += 1
    print(f"B {B} keys {nk}: {n} launches, {bad} differ from the first", flush=True)
