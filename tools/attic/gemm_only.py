#!/usr/bin/env python3
"""The decoder's four GEMM launches at the benchmark shape, a few times each - a small target for rocprofv3 --pmc passes
(tools/gemm_pmc.sh).  Inputs are random; residual / statistics / row_scale operands as in the folded inference layer."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from aki_amd import ops  # noqa: E402

dev = "cuda"
M, d, F = 8 * 655, 3072, 8192
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s, sc=1.0: (torch.randn(*s, device=dev, generator=g) * sc).to(torch.bfloat16)
h, o, a = rnd(M, d), rnd(M, d), rnd(M, F)
wo, wg, wd = rnd(d, d, sc=0.02), rnd(2 * F, d, sc=0.02), rnd(d, F, sc=0.02)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
for _ in range(n):
    st = ops.new_stats(M, dev)
    h2 = ops.linear(o, wo, residual=h, stats_out=st, stats_eps=1e-5)
    act = ops.linear(h2, wg, act=ops.ACT_SWIGLU, row_scale=st.rstd)
    st3 = ops.new_stats(M, dev)
    h3 = ops.linear(act, wd, residual=h2, stats_out=st3, stats_eps=1e-5)
    h4 = ops.linear(o, wo)            # plain, no residual / statistics: the kernel without its epilogue extras
torch.cuda.synchronize()
print("done")
