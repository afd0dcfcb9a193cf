#!/usr/bin/env python3
"""Three-deep against two-deep weight ring in the pipelined 256x256 GEMM (lab bit 2048 selects the two-deep loop), operands re-used every launch (hot: on-die)
or rotated through 12 buffers (cold: from HBM, as in the forward); outputs must be bit-identical.   python tools/weight_ring_ab.py"""
import sys; sys.path.insert(0, "/root/repo")
import torch
from aki_amd import ops, _lib
lib = _lib.load_lab(); _lib._lib = lib
dev = "cuda"; NB = 12
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s, sc=1.0: (torch.randn(*s, device=dev, generator=g) * sc).to(torch.bfloat16)
M = 8 * 655
def bench(name, N, K, act, n_out, res=False):
    xs = [rnd(M, K) for _ in range(NB)]; ws = [rnd(N, K, sc=0.02) for _ in range(NB)]
    y = torch.empty(M, n_out, device=dev, dtype=torch.bfloat16)
    rs = torch.rand(M, device=dev) + 0.5 if not res else None
    rr = rnd(M, n_out) if res else None
    st = ops.new_stats(M, dev) if res else None
    kw = dict(row_scale=rs) if not res else dict(residual=rr, stats_out=st, stats_eps=1e-5)
    def run(mode, cold, iters=2 * NB):
        lib.aki_lab_set_gemm_tile(mode)
        evs = []
        for i in range(iters):
            w = ws[i % NB if cold else 0]; x = xs[i % NB if cold else 0]
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); ops.linear(x, w, act=act, out=y, **kw); b.record(); evs.append((a, b))
        torch.cuda.synchronize()
        return sum(a.elapsed_time(b) for a, b in evs) / iters * 1e3
    ref = None
    for mode in (0, 2048):
        lib.aki_lab_set_gemm_tile(mode); yy = ops.linear(xs[0], ws[0], act=act, **kw); torch.cuda.synchronize()
        if ref is None: ref = yy.clone()
        else: assert torch.equal(ref, yy), name
    res = {}
    for mode in (0, 2048):
        for cold in (0, 1): run(mode, cold, NB)
    for mode in (0, 2048):
        for cold in (0, 1):
            res[(mode, cold)] = min(run(mode, cold) for _ in range(4))
    lib.aki_lab_set_gemm_tile(0)
    print(f"{name:22s} three-deep W ring: hot {res[(0,0)]:6.1f} cold {res[(0,1)]:6.1f}   two-deep: hot {res[(2048,0)]:6.1f} cold {res[(2048,1)]:6.1f}")
bench("gate_up + SwiGLU", 16384, 3072, ops.ACT_SWIGLU, 8192)
bench("lm_head", 32064, 3072, 0, 32064)
bench("qkv-shaped plain", 9216, 3072, 0, 9216)
bench("siglip-qkv-like K1152", 3456, 1152, 0, 3456)
bench("o_proj +res +stats", 3072, 3072, 0, 3072, res=True)
bench("down +res +stats", 3072, 8192, 0, 3072, res=True)
