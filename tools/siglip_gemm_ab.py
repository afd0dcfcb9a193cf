#!/usr/bin/env python3
"""The four GEMMs of a SigLIP encoder layer AS THE FOLDED INFERENCE LAYER LAUNCHES THEM (aki_amd/siglip.py: qkv with the folded
LayerNorm, out-proj with bias + residual + statistics, fc1 with folded LayerNorm + GELU, fc2 with bias + residual + statistics) at
the benchmark batch (8 images x 576 patches), under each tile configuration of the lab library, interleaved rounds in one process.
Also the decoder's one-round GEMMs with their epilogue pieces switched on one at a time (what residual / statistics cost).

    python tools/siglip_gemm_ab.py [--modes 0,1,2,3,4] [--rounds 5] [--iters 20]
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from aki_amd import _lib, ops  # noqa: E402

dev = "cuda"


def loop_us(fn, iters):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--modes", default="0,1,2,3,4")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--json", default=None)
    ap.add_argument("--prev", default=None, help="path of ANOTHER build of the library (e.g. aki_amd/lib/abl/libaki_prev.so): every case is "
                                                 "then timed on both builds in turn (tile modes are ignored) and the outputs must be bit-identical")
    a = ap.parse_args()
    modes = [int(m) for m in a.modes.split(",")]
    if a.prev:
        libs = {"new": _lib.load(), "prev": _lib._bind(a.prev)}
        lib = None
        modes = ["new", "prev"]
    else:
        lib = _lib.load_lab()
        _lib._lib = lib                      # route ops.* through the lab twin for this script

    def select(m):
        if a.prev:
            _lib._lib = libs[m]
        else:
            lib.aki_lab_set_gemm_tile(m)
    g = torch.Generator(device=dev).manual_seed(0)
    rnd = lambda *s, sc=1.0: (torch.randn(*s, device=dev, generator=g) * sc).to(torch.bfloat16)
    f32 = lambda *s: torch.randn(*s, device=dev, generator=g)
    M, E, F, Fp = 8 * 576, 1152, 4304, 4352
    h, att = rnd(M, E), rnd(M, E)
    st = ops.RowStats(f32(M).abs() + 0.5, f32(M) * 0.1)
    wqkv, bqkv, cqkv = rnd(3 * E, E, sc=0.03), rnd(3 * E, sc=0.1), f32(3 * E)
    wo, bo = rnd(E, E, sc=0.03), rnd(E, sc=0.1)
    w1, b1, c1 = rnd(F, E, sc=0.03), rnd(F, sc=0.1), f32(F)
    w2, b2 = rnd(E, Fp, sc=0.02), rnd(E, sc=0.1)
    hbuf = torch.zeros(M, Fp, device=dev, dtype=torch.bfloat16)
    hbuf[:, :F] = rnd(M, F)
    so = ops.new_stats(M, dev, ln=True)
    yq, yo, y2 = (torch.empty(M, n_, device=dev, dtype=torch.bfloat16) for n_ in (3 * E, E, E))
    Md, d, Fd = 8 * 655, 3072, 8192
    xd, rd, ad = rnd(Md, d), rnd(Md, d), rnd(Md, Fd)
    wod, wdd = rnd(d, d, sc=0.02), rnd(d, Fd, sc=0.02)
    sd = ops.new_stats(Md, dev)
    yd = torch.empty(Md, d, device=dev, dtype=torch.bfloat16)
    wgd, whd = rnd(2 * Fd, d, sc=0.02), rnd(32064, d, sc=0.02)
    sd2 = f32(Md).abs() + 0.5
    ygd, yhd = torch.empty(Md, Fd, device=dev, dtype=torch.bfloat16), torch.empty(Md, 32064, device=dev, dtype=torch.bfloat16)
    cases = {
        "siglip qkv (LN fold)": (lambda: ops.linear(h, wqkv, bias=bqkv, row_scale=st.rstd, row_shift=st.mean, col_shift=cqkv, out=yq), 2.0 * M * 3 * E * E),
        "siglip out (+res +stats)": (lambda: ops.linear(att, wo, bias=bo, residual=h, stats_out=so, stats_eps=1e-6, out=yo), 2.0 * M * E * E),
        "siglip fc1 (LN fold, gelu)": (lambda: ops.linear(h, w1, bias=b1, act=ops.ACT_GELU_TANH, out=hbuf[:, :F], row_scale=st.rstd, row_shift=st.mean, col_shift=c1), 2.0 * M * F * E),
        "siglip fc2 (+res +stats)": (lambda: ops.linear(hbuf, w2, bias=b2, residual=h, stats_out=so, stats_eps=1e-6, out=y2), 2.0 * M * E * Fp),
        "siglip out plain": (lambda: ops.linear(att, wo, out=yo), 2.0 * M * E * E),
        "siglip fc2 plain": (lambda: ops.linear(hbuf, w2, out=y2), 2.0 * M * E * Fp),
        "lm o_proj plain": (lambda: ops.linear(xd, wod, out=yd), 2.0 * Md * d * d),
        "lm o_proj +res": (lambda: ops.linear(xd, wod, residual=rd, out=yd), 2.0 * Md * d * d),
        "lm o_proj +stats": (lambda: ops.linear(xd, wod, stats_out=sd, stats_eps=1e-5, out=yd), 2.0 * Md * d * d),
        "lm o_proj +res +stats": (lambda: ops.linear(xd, wod, residual=rd, stats_out=sd, stats_eps=1e-5, out=yd), 2.0 * Md * d * d),
        "lm gate_up swiglu (row_scale)": (lambda: ops.linear(xd, wgd, act=ops.ACT_SWIGLU, row_scale=sd2, out=ygd), 2.0 * Md * 2 * Fd * d),
        "lm head plain": (lambda: ops.linear(xd, whd, out=yhd), 2.0 * Md * 32064 * d),
        "lm down plain": (lambda: ops.linear(ad, wdd, out=yd), 2.0 * Md * d * Fd),
        "lm down +res": (lambda: ops.linear(ad, wdd, residual=rd, out=yd), 2.0 * Md * d * Fd),
        "lm down +res +stats": (lambda: ops.linear(ad, wdd, residual=rd, stats_out=sd, stats_eps=1e-5, out=yd), 2.0 * Md * d * Fd),
    }
    out = []
    for name, (fn, fl) in cases.items():
        ms_ = modes if (name.startswith("siglip") or a.prev) else [0]
        ref = None
        t = {m: [] for m in ms_}
        for m in ms_:
            select(m)
            for _ in range(3):
                y = fn()
            torch.cuda.synchronize()
            if ref is None:
                ref = y.clone()
            else:
                assert torch.equal(ref, y), f"{name}: tile mode {m} changes the result"
        for _ in range(a.rounds):
            for m in ms_:
                select(m)
                t[m].append(loop_us(fn, a.iters))
        select(ms_[0] if a.prev else 0)
        rec = {"case": name}
        for m in ms_:
            v = sorted(t[m])
            rec[f"mode{m}" if not a.prev else m] = {"min_us": round(v[0], 1), "med_us": round(v[len(v) // 2], 1), "tflops_med": round(fl / v[len(v) // 2] / 1e6, 1)}
        out.append(rec)
        print(json.dumps(rec), flush=True)
    if a.json:
        json.dump(out, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
