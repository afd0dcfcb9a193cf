"""Split-K workgroup order, A/B in one process (lab library): slice-major (an XCD reads one or two K slices of the activation) against tile-major
(the slices of a tile are neighbours), the planner's split launches of a one-sample prefill, cold operands.   python tools/attic/slice_major_ab.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from aki_amd import _lib, ops
dev = "cuda"
lib = _lib.load_lab(); _lib._lib = lib
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s, sc=1.0: (torch.randn(*s, device=dev, generator=g) * sc).to(torch.bfloat16)
def timed(fn, NB=12, reps=4):
    for i in range(NB): fn(i)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for k in range(reps * NB): fn(k % NB)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / (reps * NB) * 1e3)
    return best
for (M, N, K, ln, name) in ((655, 3072, 3072, False, "o_proj M655"), (655, 3072, 8192, False, "down M655"), (207, 3072, 3072, False, "o_proj M207"), (207, 3072, 8192, False, "down M207"),
                            (576, 1152, 4352, True, "siglip fc2 M576")):
    x = [rnd(M, K) for _ in range(12)]; w = [rnd(N, K, sc=0.02) for _ in range(12)]; r = [rnd(M, N) for _ in range(12)]
    y = torch.empty(M, N, device=dev, dtype=torch.bfloat16); st = ops.new_stats(M, dev, ln=ln)
    fn = lambda i: ops.linear(x[i], w[i], residual=r[i], stats_out=st, stats_eps=1e-5, out=y)
    res, outs = {}, {}
    for rnd_ in range(2):
        for mode in (1, 0):
            lib.aki_lab_set_slice_major(mode)
            fn(0); outs[mode] = y.clone()
            res.setdefault(mode, []).append(timed(fn))
    lib.aki_lab_set_slice_major(0)
    print(f"{name:18s} slice-major {min(res[1]):6.1f} us   tile-major {min(res[0]):6.1f} us   outputs equal {torch.equal(outs[0], outs[1])}")
