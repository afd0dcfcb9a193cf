#!/usr/bin/env python3
"""The decoder's GEMMs at the row count of a ONE-sample prefill (M = 655: what `generate` runs before its first token) under each tile
configuration of the lab library (0 = the cost model's choice), operands rotated through 12 buffers (cold, as in the forward).
    python tools/prefill_gemm_ab.py [--M 655] [--modes 0,1,2,3,5]"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aki_amd import _lib, ops

dev = "cuda"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--M", type=int, default=655)
    ap.add_argument("--modes", default="0,1,2,3,5")
    ap.add_argument("--rounds", type=int, default=3)
    a = ap.parse_args()
    modes = [int(m) for m in a.modes.split(",")]
    lib = _lib.load_lab()
    _lib._lib = lib
    g = torch.Generator(device=dev).manual_seed(0)
    rnd = lambda *s, sc=1.0: (torch.randn(*s, device=dev, generator=g) * sc).to(torch.bfloat16)
    M, d, F = a.M, 3072, 8192
    NB = 12
    x, r, act = [rnd(M, d) for _ in range(NB)], [rnd(M, d) for _ in range(NB)], [rnd(M, F) for _ in range(NB)]
    wo, wd, wg = [rnd(d, d, sc=0.02) for _ in range(NB)], [rnd(d, F, sc=0.02) for _ in range(NB)], [rnd(2 * F, d, sc=0.02) for _ in range(NB)]
    y, yg = torch.empty(M, d, device=dev, dtype=torch.bfloat16), torch.empty(M, F, device=dev, dtype=torch.bfloat16)
    st = ops.new_stats(M, dev)
    rs = torch.rand(M, device=dev) + 0.5
    it = [0]
    cases = {
        "o_proj  N3072 K3072 (+res +stats)": (lambda i: ops.linear(x[i], wo[i], residual=r[i], stats_out=st, stats_eps=1e-5, out=y), 2.0 * M * d * d),
        "down    N3072 K8192 (+res +stats)": (lambda i: ops.linear(act[i], wd[i], residual=r[i], stats_out=st, stats_eps=1e-5, out=y), 2.0 * M * d * F),
        "gate_up N16384 K3072 (fold, swiglu)": (lambda i: ops.linear(x[i], wg[i], act=ops.ACT_SWIGLU, row_scale=rs, out=yg), 2.0 * M * 2 * F * d),
    }
    for name, (fn, fl) in cases.items():
        res = {}
        for _ in range(a.rounds):
            for m in modes:
                lib.aki_lab_set_gemm_tile(m)
                for i in range(NB):
                    fn(i)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for k in range(4 * NB):
                    fn(k % NB)
                e1.record()
                torch.cuda.synchronize()
                res.setdefault(m, []).append(e0.elapsed_time(e1) / (4 * NB) * 1e3)
        lib.aki_lab_set_gemm_tile(0)
        print(name, json.dumps({f"mode{m}": {"us": round(min(v), 1), "TF/s": round(fl / min(v) / 1e6, 0)} for m, v in res.items()}), flush=True)


if __name__ == "__main__":
    main()
