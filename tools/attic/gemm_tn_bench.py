#!/usr/bin/env python3
"""Weight-gradient GEMM dW = dY^T X on the operands as they lie (aki_gemm_tn) against two transposes + the forward GEMM, at the decoder's shapes.
    python tools/gemm_tn_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aki_amd import ops, train_ops as T
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s: torch.randn(*s, device=dev, generator=g).to(torch.bfloat16)
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for name, M, N, K in (("qkv", 5240, 9216, 3072), ("o_proj", 5240, 3072, 3072), ("gate_up", 5240, 16384, 3072), ("down", 5240, 3072, 8192), ("small", 300, 264, 136)):
    dy, x = rnd(M, N), rnd(M, K)
    ref = dy.float().t() @ x.float()
    got = T.gemm_tn(dy, x)
    old = ops.linear(T.transpose(dy), T.transpose(x))
    e_new = ((got.float() - ref).abs().max() / ref.abs().max()).item(); e_old = ((old.float() - ref).abs().max() / ref.abs().max()).item()
    t_new = timeit(lambda: T.gemm_tn(dy, x)); t_old = timeit(lambda: ops.linear(T.transpose(dy), T.transpose(x))); t_mm = timeit(lambda: ops.linear(old, old[:, :0].new_zeros(8, old.shape[1]))) if False else 0
    dyT, xT = T.transpose(dy), T.transpose(x)
    t_gemm = timeit(lambda: ops.linear(dyT, xT))
    fl = 2.0 * M * N * K
    print(f"{name:8s} dW [{N} x {K}] over {M} rows: gemm_tn {t_new:7.1f} us ({fl / t_new / 1e6:6.0f} TF/s, err {e_new:.1e})   transposes + GEMM {t_old:7.1f} us (GEMM alone {t_gemm:7.1f}, err {e_old:.1e})   equal: {torch.equal(got, old)}")
