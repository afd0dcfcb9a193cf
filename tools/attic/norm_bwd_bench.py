#!/usr/bin/env python3
"""RMSNorm / LayerNorm backward and the bias-gradient column sum at the training shapes (two launches each: the row pass and the fold of its partial rows).
    python tools/norm_bwd_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aki_amd import train_ops as T
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s: torch.randn(*s, device=dev, generator=g).to(torch.bfloat16)
def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for rows, cols, rms in ((5240, 3072, True), (4608, 1152, False), (1152, 1152, False)):
    x, dy, w = rnd(rows, cols), rnd(rows, cols), rnd(cols)
    us = timeit(lambda: T.norm_bwd(rms, x, w, dy, 1e-5, need_db=not rms))
    dr = rnd(rows, cols)
    us_r = timeit(lambda: T.norm_bwd(rms, x, w, dy, 1e-5, need_db=not rms, dres=dr))
    dx, dw, db = T.norm_bwd(rms, x, w, dy, 1e-5, need_db=not rms)
    xf, dyf, wf = x.float().requires_grad_(True), dy.float(), w.float().requires_grad_(True)
    y = (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-5) * wf) if rms else torch.nn.functional.layer_norm(xf, (cols,), wf, torch.zeros_like(wf), 1e-5)
    y.backward(dyf)
    err = (dw.float() - wf.grad).abs().max().item() / wf.grad.abs().max().item()
    print(f"norm_bwd {'rms' if rms else 'ln '} {rows} x {cols}: {us:7.1f} us ({us_r:7.1f} with the residual-branch gradient)   dw max err / max |dw| = {err:.2e}")
x = rnd(5240, 9216)
print(f"colsum 5240 x 9216: {timeit(lambda: T.colsum(x)):7.1f} us   err {((T.colsum(x).float() - x.float().sum(0)).abs().max() / x.float().sum(0).abs().max()).item():.2e}")
