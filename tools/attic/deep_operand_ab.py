#!/usr/bin/env python3
"""Which operand of the residual GEMMs (o_proj, down) should sit on the three-deep LDS ring: weights (lab bit 13) or tokens (lab bit 12)?
Operands rotated through 12 buffers (cold: from HBM, as in the forward) or re-used (hot).   python tools/deep_operand_ab.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aki_amd import ops, _lib
lib = _lib.load_lab(); _lib._lib = lib
dev = "cuda"; NB = 12
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s, sc=1.0: (torch.randn(*s, device=dev, generator=g) * sc).to(torch.bfloat16)
M = 8 * 655
def bench(name, N, K):
    xs = [rnd(M, K) for _ in range(NB)]; ws = [rnd(N, K, sc=0.02) for _ in range(NB)]
    rr = rnd(M, N); st = ops.new_stats(M, dev); y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    def run(mode, cold, iters=2 * NB):
        lib.aki_lab_set_gemm_tile(mode)
        evs = []
        for i in range(iters):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); ops.linear(xs[i % NB if cold else 0], ws[i % NB if cold else 0], residual=rr, stats_out=st, stats_eps=1e-5, out=y); b.record(); evs.append((a, b))
        torch.cuda.synchronize()
        return sum(a.elapsed_time(b) for a, b in evs) / iters * 1e3
    modes = {8192: "weights deep", 4096: "tokens deep", 2048: "two-deep (residual in two stages)", 0: "product choice"}
    ref = None
    for m in modes:
        lib.aki_lab_set_gemm_tile(m); yy = ops.linear(xs[0], ws[0], residual=rr, stats_out=st, stats_eps=1e-5).clone(); torch.cuda.synchronize()
        if ref is None: ref = yy
        else: assert torch.equal(ref, yy), (name, m)
    for m in modes:
        for c in (0, 1): run(m, c, NB)
    out = {}
    for rep in range(4):
        for m in modes:
            for c in (0, 1):
                out.setdefault((m, c), []).append(run(m, c))
    lib.aki_lab_set_gemm_tile(0)
    print(name + ": " + "; ".join(f"{v}: hot {min(out[(m, 0)]):6.1f} cold {min(out[(m, 1)]):6.1f}" for m, v in modes.items()))
bench("o_proj +res +stats", 3072, 3072)
bench("down +res +stats", 3072, 8192)
