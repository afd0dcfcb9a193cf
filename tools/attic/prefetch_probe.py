#!/usr/bin/env python3
"""Would prefetching the NEXT launch's weight matrix pay?  The decoder's GEMMs read weights that were last touched one
layer (≈2.4 GB of traffic) ago: they come from HBM, and the single-round launches (down, o_proj: one tile per CU, every CU
starts on a cold panel at once) run 10-18 % slower in the step than back to back on one resident weight.  This probe times
the GEMM alone (HIP events around it) on rotating weights in three states:
  cold            nothing touches the weight beforehand
  touched         a streaming read of the weight (what an in-kernel prefetch from the previous launch would do) right before
  touched+traffic the same, followed by an unrelated gate_up-sized GEMM (≈870 MB of fabric traffic) before the timed launch
and on ONE weight back to back (hot).  If `touched+traffic` stays near `hot`, the 256 MB Infinity Cache keeps a prefetched
matrix through a whole launch of other traffic and an in-kernel prefetch is worth building.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from aki_amd import ops  # noqa: E402

dev = "cuda"
M, NW = 8 * 655, 10
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s, sc=1.0: (torch.randn(*s, device=dev, generator=g) * sc).to(torch.bfloat16)
xf, wf = rnd(M, 3072), rnd(16384, 3072, sc=0.02)         # the unrelated traffic: a gate_up launch


def timed(fn):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    fn()
    b.record()
    return a, b


for name, N, K in [("down", 3072, 8192), ("o_proj", 3072, 3072), ("qkv-shaped", 9216, 3072)]:
    x = rnd(M, K)
    res = rnd(M, N)
    ws = [rnd(N, K, sc=0.02) for _ in range(NW)]
    out = {}
    for mode in ("hot", "cold", "touched", "touched+traffic"):
        evs = []
        for rep in range(3 * NW):
            w = ws[0] if mode == "hot" else ws[rep % NW]
            if mode.startswith("touched"):
                w.view(torch.int64).sum()
            if mode == "touched+traffic":
                ops.linear(xf, wf, act=ops.ACT_SWIGLU)
            evs.append(timed(lambda: ops.linear(x, w, residual=res)))
        torch.cuda.synchronize()
        ts = sorted(a.elapsed_time(b) * 1e3 for a, b in evs[NW:])
        out[mode] = ts[len(ts) // 2]
    print(f"{name:12s} N{N} K{K} ({N * K * 2 / 2**20:.0f} MiB weight): " + "  ".join(f"{k} {v:6.1f} us" for k, v in out.items()), flush=True)
