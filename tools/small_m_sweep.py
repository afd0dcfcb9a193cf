#!/usr/bin/env python3
"""Sweep of the small-M tile variants (gemm_bf16.hip: launch_variant - tile shape x ring depth) and K splits over the GEMM shapes of a
ONE-sample prefill: the decoder at M = 655 / 207 and the SigLIP tower at one 336 px image (M = 576), launched as the model launches them,
cold operands (12 rotating buffers).  Lab library: every bf16 GEMM is forced to (variant, ksplit).  Each configuration is also checked:
equal to the planner's output bit for bit when ksplit == 1 (same K order per output element), within bf16 rounding of it otherwise, and
bit-identical between two runs (the split-K fold adds in slice order whatever the arrival order).
    python tools/small_m_sweep.py [--M 655,207] [--siglip] [--variants 0,1,...] [--splits 1,2,3,4,6,8]"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aki_amd import _lib, ops

dev = "cuda"
NB = 12


def timed(fn, reps=3):
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
    ap.add_argument("--M", default="655,207")
    ap.add_argument("--siglip", action="store_true")
    ap.add_argument("--siglip-m", type=int, default=576, help="rows of the SigLIP cases: 576 = one 336 px image, 4608 = the headline batch of 8")
    ap.add_argument("--variants", default="0,1,2,3,4,5,6,7,8,9")
    ap.add_argument("--splits", default="1,2,3,4,6,8")
    ap.add_argument("--only", default="", help="substring filter on the case names")
    a = ap.parse_args()
    lib = _lib.load_lab()
    _lib._lib = lib
    ops.SPLITK_WS_MIN_BYTES = 512 << 20
    ops.SPLITK_MAX_M = 1 << 20
    variants = [int(v) for v in a.variants.split(",")]
    splits = [int(v) for v in a.splits.split(",")]
    g = torch.Generator(device=dev).manual_seed(0)
    rnd = lambda *s, sc=1.0: (torch.randn(*s, device=dev, generator=g) * sc).to(torch.bfloat16)
    d, F, H = 3072, 8192, 32
    cases = []          # (name, fn(i) -> output tensor, can_split, flops)
    keep = []
    if a.M:
        wq, wo = [rnd(3 * d, d, sc=0.02) for _ in range(NB)], [rnd(d, d, sc=0.02) for _ in range(NB)]
        wg, wd = [rnd(2 * F, d, sc=0.02) for _ in range(NB)], [rnd(d, F, sc=0.02) for _ in range(NB)]
        for M in [int(m) for m in a.M.split(",")]:
            x, r, act = [rnd(M, d) for _ in range(NB)], [rnd(M, d) for _ in range(NB)], [rnd(M, F) for _ in range(NB)]
            o_in = [t_.view(1, M, d) for t_ in x]
            kc, vc = torch.zeros(1, H, M + 64, 96, device=dev, dtype=torch.bfloat16), torch.zeros(1, H, M + 64, 96, device=dev, dtype=torch.bfloat16)
            pos = torch.arange(M + 64, device=dev, dtype=torch.float32)[:, None] * torch.arange(96, device=dev, dtype=torch.float32)[None, :] * 1e-3
            cos, sin = pos.cos().contiguous(), pos.sin().contiguous()
            y, yg = torch.empty(M, d, device=dev, dtype=torch.bfloat16), torch.empty(M, F, device=dev, dtype=torch.bfloat16)
            st = ops.new_stats(M, dev)
            rs = torch.rand(M, device=dev) + 0.5
            keep.append((x, r, act, kc, vc, cos, sin, y, yg, st, rs))

            def mk(M=M, x=x, r=r, act=act, o_in=o_in, kc=kc, vc=vc, cos=cos, sin=sin, y=y, yg=yg, st=st, rs=rs):
                qb = [None]
                def qkv(i):
                    qb[0] = ops.qkv_rope(o_in[i], wq[i], cos, sin, H, k_out=kc, v_out=vc, row_scale=rs)[0]
                def oproj(i):
                    ops.linear(x[i], wo[i], residual=r[i], stats_out=st, stats_eps=1e-5, out=y)
                def gateup(i):
                    ops.linear(x[i], wg[i], act=ops.ACT_SWIGLU, row_scale=rs, out=yg)
                def down(i):
                    ops.linear(act[i], wd[i], residual=r[i], stats_out=st, stats_eps=1e-5, out=y)
                o_q = lambda: torch.cat([qb[0].flatten().float(), kc.flatten().float(), vc.flatten().float()])
                o_y = lambda: torch.cat([y.flatten().float(), st.rstd])
                return [(f"M{M} qkv+rope N9216 K3072", qkv, o_q, False, 2.0 * M * 3 * d * d), (f"M{M} o_proj N3072 K3072", oproj, o_y, True, 2.0 * M * d * d),
                        (f"M{M} gate_up N16384 K3072", gateup, lambda: yg.float(), False, 2.0 * M * 2 * F * d), (f"M{M} down N3072 K8192", down, o_y, True, 2.0 * M * d * F)]
            cases += mk()
    if a.siglip:
        M, E, I, Ip = a.siglip_m, 1152, 4304, 4352
        x, r = [rnd(M, E) for _ in range(NB)], [rnd(M, E) for _ in range(NB)]
        a1 = [torch.zeros(M, Ip, device=dev, dtype=torch.bfloat16) for _ in range(NB)]
        for t_ in a1:
            t_[:, :I] = rnd(M, I)
        wqkv, wout = [rnd(3 * E, E, sc=0.03) for _ in range(NB)], [rnd(E, E, sc=0.03) for _ in range(NB)]
        w1 = [rnd(I, E, sc=0.03) for _ in range(NB)]
        w2 = [torch.zeros(E, Ip, device=dev, dtype=torch.bfloat16) for _ in range(NB)]
        for t_ in w2:
            t_[:, :I] = rnd(E, I, sc=0.03)
        b3, b1, bE = rnd(3 * E), rnd(I), rnd(E)
        c3, c1 = torch.randn(3 * E, device=dev), torch.randn((I + 3) // 4 * 4, device=dev)
        rs, mu = torch.rand(M, device=dev) + 0.5, torch.randn(M, device=dev) * 0.1
        st = ops.new_stats(M, dev, ln=True)
        y, y3, y1 = torch.empty(M, E, device=dev, dtype=torch.bfloat16), torch.empty(M, 3 * E, device=dev, dtype=torch.bfloat16), torch.empty(M, Ip, device=dev, dtype=torch.bfloat16)

        def sq(i):
            ops.linear(x[i], wqkv[i], bias=b3, row_scale=rs, row_shift=mu, col_shift=c3, out=y3)
        def so(i):
            ops.linear(x[i], wout[i], bias=bE, residual=r[i], stats_out=st, stats_eps=1e-6, out=y)
        def s1(i):
            ops.linear(x[i], w1[i], bias=b1, act=ops.ACT_GELU_TANH, row_scale=rs, row_shift=mu, col_shift=c1, out=y1[:, :I])
        def s2(i):
            ops.linear(a1[i], w2[i], bias=bE, residual=r[i], stats_out=st, stats_eps=1e-6, out=y)
        o_ys = lambda: torch.cat([y.flatten().float(), st.rstd, st.mean])
        cases += [(f"siglip M{M} qkv N3456 K1152", sq, lambda: y3.float(), False, 2.0 * M * 3 * E * E), (f"siglip M{M} out N1152 K1152", so, o_ys, True, 2.0 * M * E * E),
                  (f"siglip M{M} fc1 N4304 K1152", s1, lambda: y1[:, :I].float(), False, 2.0 * M * I * E), (f"siglip M{M} fc2 N1152 K4352", s2, o_ys, True, 2.0 * M * E * Ip)]
    if a.only:
        cases = [c for c in cases if a.only in c[0]]
    print(f"# {torch.cuda.get_device_name(0)}; cold operands; us per launch (TF/s); '!' = output differs from the planner's beyond bf16 rounding, "
          f"'~' = not bit-reproducible")
    for name, fn, out, can_split, fl in cases:
        lib.aki_lab_set_small_m(-1, 1)
        fn(0)
        ref = out().clone()
        base = timed(fn)
        row = {"planner": round(base, 1)}
        best = ("planner", base)
        for v in variants:
            for ks in (splits if can_split else [1]):
                lib.aki_lab_set_small_m(v, ks)
                try:
                    fn(0)
                    o1 = out().clone()
                    fn(0)
                    o2 = out().clone()
                except Exception as e:                       # a variant the epilogue does not support
                    row[f"v{v}k{ks}"] = "n/a"
                    continue
                torch.cuda.synchronize()
                flag = ""
                err = (o1 - ref).abs().max().item()
                tol = 2.0 ** -7 * max(1.0, ref.abs().max().item())
                if (ks == 1 and err != 0.0 and err > tol) or (ks > 1 and err > tol) or err != err:
                    flag += "!"
                if not torch.equal(o1, o2):
                    flag += "~"
                us = timed(fn)
                row[f"v{v}k{ks}"] = f"{us:.1f}{flag}"
                if not flag and us < best[1]:
                    best = (f"v{v}k{ks}", us)
        lib.aki_lab_set_small_m(-1, 1)
        print(f"{name}: best {best[0]} {best[1]:.1f} us ({fl / best[1] / 1e6:.0f} TF/s) vs planner {base:.1f} us  {json.dumps(row)}", flush=True)


if __name__ == "__main__":
    main()
