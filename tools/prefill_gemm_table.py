#!/usr/bin/env python3
"""The four decoder GEMMs (and the head) at the row counts of a ONE-sample prefill - M = 655 (1 image + 512-token prompt: what the
reference's callers run, local_demo.py:75-87, eval_cv_bench/eval.py:92-104) and M = 207 (BASELINE configs[0]) - as the model launches them
(folded RMSNorm: row_scale on qkv / gate_up, residual + row statistics on o_proj / down), operands rotated through 12 buffers (cold, as in
the forward: weights come out of HBM in every launch).  Per GEMM: us, TF/s, TB/s of weight bytes; per M: the layer's total against the
two floors (FLOPs at 1.3 PF/s sustained, weight bytes at 5 TB/s).
    python tools/prefill_gemm_table.py [--M 207,655] [--lab MODE]      -> profiles/r05_prefill_gemm_table.txt"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aki_amd import _lib, ops

dev = "cuda"


def timed(fn, NB, reps=4):
    for i in range(NB):
        fn(i)
    torch.cuda.synchronize()
    best = 1e30
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for k in range(reps * NB):
            fn(k % NB)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / (reps * NB) * 1e3)
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--M", default="207,655")
    ap.add_argument("--lab", type=int, default=None, help="route through the lab library with this forced tile mode")
    ap.add_argument("--head", action="store_true", help="also the lm_head at M rows (generate only needs the last row)")
    a = ap.parse_args()
    if a.lab is not None:
        lib = _lib.load_lab()
        _lib._lib = lib
        lib.aki_lab_set_gemm_tile(a.lab)
    g = torch.Generator(device=dev).manual_seed(0)
    rnd = lambda *s, sc=1.0: (torch.randn(*s, device=dev, generator=g) * sc).to(torch.bfloat16)
    d, F, H, V = 3072, 8192, 32, 32064
    NB = 12
    wq, wo = [rnd(3 * d, d, sc=0.02) for _ in range(NB)], [rnd(d, d, sc=0.02) for _ in range(NB)]
    wg, wd = [rnd(2 * F, d, sc=0.02) for _ in range(NB)], [rnd(d, F, sc=0.02) for _ in range(NB)]
    wh = [rnd(V, d, sc=0.02) for _ in range(4)] if a.head else None
    print(f"# {torch.cuda.get_device_name(0)}; cold operands ({NB} rotating buffers); lab mode {a.lab}")
    print(f"{'M':>5} {'GEMM':<34} {'us':>8} {'TF/s':>7} {'TB/s(w)':>8}")
    for M in [int(m) for m in a.M.split(",")]:
        x, r, act = [rnd(M, d) for _ in range(NB)], [rnd(M, d) for _ in range(NB)], [rnd(M, F) for _ in range(NB)]
        o_in = [rnd(1, M, d) for _ in range(NB)]
        kc, vc = torch.empty(1, H, M + 64, 96, device=dev, dtype=torch.bfloat16), torch.empty(1, H, M + 64, 96, device=dev, dtype=torch.bfloat16)
        pos = torch.arange(M + 64, device=dev, dtype=torch.float32)[:, None] * torch.arange(96, device=dev, dtype=torch.float32)[None, :] * 1e-3
        cos, sin = pos.cos().contiguous(), pos.sin().contiguous()
        y, yg = torch.empty(M, d, device=dev, dtype=torch.bfloat16), torch.empty(M, F, device=dev, dtype=torch.bfloat16)
        st = ops.new_stats(M, dev)
        rs = torch.rand(M, device=dev) + 0.5
        cases = [
            ("qkv+rope N9216 K3072 (fold)", lambda i: ops.qkv_rope(o_in[i], wq[i], cos, sin, H, k_out=kc, v_out=vc, row_scale=rs), 3 * d, d),
            ("o_proj   N3072 K3072 (+res +stats)", lambda i: ops.linear(x[i], wo[i], residual=r[i], stats_out=st, stats_eps=1e-5, out=y), d, d),
            ("gate_up  N16384 K3072 (fold, swiglu)", lambda i: ops.linear(x[i], wg[i], act=ops.ACT_SWIGLU, row_scale=rs, out=yg), 2 * F, d),
            ("down     N3072 K8192 (+res +stats)", lambda i: ops.linear(act[i], wd[i], residual=r[i], stats_out=st, stats_eps=1e-5, out=y), d, F),
        ]
        if a.head:
            yh = torch.empty(M, V, device=dev, dtype=torch.bfloat16)
            cases.append(("lm_head  N32064 K3072 (fold)", lambda i: ops.linear(x[i], wh[i % 4], row_scale=rs, out=yh), V, d))
        tot_us = tot_fl = tot_wb = 0.0
        for name, fn, N, K in cases:
            us = timed(fn, NB)
            fl, wb = 2.0 * M * N * K, 2.0 * N * K
            print(f"{M:>5} {name:<34} {us:8.1f} {fl / us / 1e6:7.0f} {wb / us / 1e6:8.2f}", flush=True)
            if not name.startswith("lm_head"):
                tot_us, tot_fl, tot_wb = tot_us + us, tot_fl + fl, tot_wb + wb
        print(f"{M:>5} {'layer (4 GEMMs)':<34} {tot_us:8.1f} {tot_fl / tot_us / 1e6:7.0f} {tot_wb / tot_us / 1e6:8.2f}   floors: "
              f"{tot_fl / 1.3e9:.0f} us (1.3 PF/s), {tot_wb / 5e6:.0f} us (5 TB/s); x32 layers = {tot_us * 32 / 1e3:.2f} ms")


if __name__ == "__main__":
    main()
