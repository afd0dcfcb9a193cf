#!/usr/bin/env python3
"""SigLIP's four GEMMs (as the folded layer launches them) with their operands re-used (hot) or rotated through 12 buffers (cold, as in the forward:
820 MB of weights per step, fc2's 40 MB activation).   python tools/siglip_cold_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aki_amd import ops, _lib
lib = _lib.load_lab(); _lib._lib = lib
dev = "cuda"; NB = 12
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s, sc=1.0: (torch.randn(*s, device=dev, generator=g) * sc).to(torch.bfloat16)
f32 = lambda *s: torch.randn(*s, device=dev, generator=g)
M, E, F, Fp = 8 * 576, 1152, 4304, 4352
def bench(name, N, K, mk):
    xs = [rnd(M, K) for _ in range(NB)]; ws = [rnd(N, K, sc=0.03) for _ in range(NB)]
    fn = mk()
    def run(cw, cx, iters=2 * NB):
        evs = []
        for i in range(iters):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); fn(xs[i % NB if cx else 0], ws[i % NB if cw else 0]); b.record(); evs.append((a, b))
        torch.cuda.synchronize()
        return sum(a.elapsed_time(b) for a, b in evs) / iters * 1e3
    for mode, tag in ((0, "product"), (2048, "two-stage loops")):
        lib.aki_lab_set_gemm_tile(mode)
        for c in ((0, 0), (1, 0), (0, 1), (1, 1)): run(*c, NB)
        r = {c: min(run(*c) for _ in range(4)) for c in ((0, 0), (1, 0), (0, 1), (1, 1))}
        print(f"{name:28s} {tag:16s} hot {r[(0,0)]:6.1f}  W cold {r[(1,0)]:6.1f}  X cold {r[(0,1)]:6.1f}  both cold {r[(1,1)]:6.1f}")
    lib.aki_lab_set_gemm_tile(0)
st = ops.RowStats(f32(M).abs() + 0.5, f32(M) * 0.1)
h = rnd(M, E); so = ops.new_stats(M, dev, ln=True)
b3, c3 = rnd(3 * E, sc=0.1), f32(3 * E)
bench("qkv (LN fold)", 3 * E, E, lambda: (lambda x, w: ops.linear(x, w, bias=b3, row_scale=st.rstd, row_shift=st.mean, col_shift=c3)))
bo = rnd(E, sc=0.1)
bench("out (+bias +res +stats)", E, E, lambda: (lambda x, w: ops.linear(x, w, bias=bo, residual=h, stats_out=so, stats_eps=1e-6)))
b1, c1 = rnd(F, sc=0.1), f32(F)
bench("fc1 (LN fold, gelu)", F, E, lambda: (lambda x, w: ops.linear(x, w, bias=b1, act=ops.ACT_GELU_TANH, row_scale=st.rstd, row_shift=st.mean, col_shift=c1)))
bench("fc2 (+bias +res +stats)", E, Fp, lambda: (lambda x, w: ops.linear(x, w, bias=bo, residual=h, stats_out=so, stats_eps=1e-6)))
