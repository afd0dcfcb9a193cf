#!/usr/bin/env python3
"""tools/stats_stress.py under each tile / loop configuration of the lab library (different timing of the same hand-off): every launch's output and statistics
must be bit-identical with the first launch of its case.   python tools/stats_stress_modes.py [--seconds 20]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aki_amd import ops, _lib
ap = argparse.ArgumentParser(); ap.add_argument("--seconds", type=float, default=20.0); a = ap.parse_args()
lib = _lib.load_lab(); _lib._lib = lib
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
shapes = [(5240, 3072, 3072), (5240, 3072, 8192), (4608, 1152, 1152), (4608, 1152, 4352), (1380, 3072, 256)]
data = {s: (torch.randn(s[0], s[2], device=dev, generator=g).to(torch.bfloat16), (torch.randn(s[1], s[2], device=dev, generator=g) * 0.05).to(torch.bfloat16),
            (torch.randn(s[0], s[1], device=dev, generator=g) * 2).to(torch.bfloat16)) for s in shapes}
modes = {0: "product", 1: "256x256", 2: "128x128", 3: "128x96", 5: "64x64 ring", 2048: "two-stage loops", 3 + 2048: "128x96 two-stage", 4096: "tokens deep", 8192: "weights deep", 1 + 4096: "256x256 tokens deep", 1 + 8192: "256x256 weights deep"}
for mode, name in modes.items():
    lib.aki_lab_set_gemm_tile(mode)
    want = {}; n = bad = 0; t0 = time.time(); per = {}
    while time.time() - t0 < a.seconds:
        for s, (x, w, r) in data.items():
            for ln in (False, True):
                outs = []
                for _ in range(4):
                    st = ops.new_stats(s[0], dev, ln=ln)
                    outs.append((ops.linear(x, w, residual=r, stats_out=st, stats_eps=1e-6), st))
                for y, st in outs:
                    key = (s, ln); n += 1
                    if key not in want: want[key] = (y.clone(), st.rstd.clone(), None if st.mean is None else st.mean.clone())
                    else:
                        y0, r0, m0 = want[key]
                        if not (torch.equal(y, y0) and torch.equal(st.rstd, r0) and (m0 is None or torch.equal(st.mean, m0))):
                            bad += 1; per[key] = per.get(key, 0) + 1
    print(f"{name:28s} {n:8d} launches, {bad} differ from the first of their case", per if per else "", flush=True)
lib.aki_lab_set_gemm_tile(0)
