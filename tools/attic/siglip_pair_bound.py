#!/usr/bin/env python3
"""Upper bound of what chaining two SigLIP GEMMs in one launch could gain (cross-GEMM dataflow, tiles of GEMM k+1
starting as their input panel completes).  A dataflow pair can at best overlap the two GEMMs completely; so time each pair of the folded
inference layer (a) back to back on one stream and (b) on TWO streams with NO dependency between them (wrong as a model, the ceiling of any
chain), both as hipGraph replays of 20 pairs (the two-stream form: 20 of one GEMM beside 20 of the other, one fork and one join), alternating in one process.  The gate was: adopt only if the pair gains >= 8 %.
    python tools/siglip_pair_bound.py [--rounds 5]"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aki_amd import ops

dev = "cuda"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--pairs", type=int, default=20)
    a = ap.parse_args()
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
    so, so2 = ops.new_stats(M, dev, ln=True), ops.new_stats(M, dev, ln=True)
    yq, yo, y2 = (torch.empty(M, n_, device=dev, dtype=torch.bfloat16) for n_ in (3 * E, E, E))
    h1 = torch.empty(M, Fp, device=dev, dtype=torch.bfloat16)
    qkv = lambda: ops.linear(h, wqkv, bias=bqkv, row_scale=st.rstd, row_shift=st.mean, col_shift=cqkv, out=yq)
    out = lambda: ops.linear(att, wo, bias=bo, residual=h, stats_out=so, stats_eps=1e-6, out=yo)
    fc1 = lambda: ops.linear(h, w1, bias=b1, act=ops.ACT_GELU_TANH, out=h1[:, :F], row_scale=st.rstd, row_shift=st.mean, col_shift=c1)
    fc2 = lambda: ops.linear(hbuf, w2, bias=b2, residual=h, stats_out=so2, stats_eps=1e-6, out=y2)
    pairs = {"out-proj -> fc1": (out, fc1), "fc1 -> fc2": (fc1, fc2), "fc2 -> qkv (next layer)": (fc2, qkv)}
    for f in (qkv, out, fc1, fc2):
        f()
    torch.cuda.synchronize()

    def graph_of(fa, fb, two_streams):
        gph = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        with torch.cuda.graph(gph):
            cur = torch.cuda.current_stream()
            if two_streams:          # ONE fork and ONE join: a fork / join per pair costs graph-edge time that a chain would not pay
                side.wait_stream(cur)
                with torch.cuda.stream(side):
                    for _ in range(a.pairs):
                        fb()
                for _ in range(a.pairs):
                    fa()
                cur.wait_stream(side)
            else:
                for _ in range(a.pairs):
                    fa()
                    fb()
        return gph

    def t_us(gph):
        gph.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            gph.replay()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / (5 * a.pairs)

    res = {}
    for name, (fa, fb) in pairs.items():
        gs, gp = graph_of(fa, fb, False), graph_of(fa, fb, True)
        seq, par = [], []
        for _ in range(a.rounds):
            seq.append(t_us(gs))
            par.append(t_us(gp))
        s_, p_ = sorted(seq)[len(seq) // 2], sorted(par)[len(par) // 2]
        res[name] = {"back_to_back_us": round(s_, 2), "two_streams_no_dependency_us": round(p_, 2), "ceiling_of_a_chain_pct": round(100 * (1 - p_ / s_), 1)}
        print(name, json.dumps(res[name]), flush=True)
    tot_s = sum(v["back_to_back_us"] for k, v in res.items() if k != "fc1 -> fc2")
    tot_p = sum(v["two_streams_no_dependency_us"] for k, v in res.items() if k != "fc1 -> fc2")
    print(json.dumps({"layer_gemms_back_to_back_us": round(tot_s, 2), "layer_gemms_if_both_pairs_overlapped_perfectly_us": round(tot_p, 2),
                      "x27_layers_ms": [round(tot_s * 27e-3, 3), round(tot_p * 27e-3, 3)]}))


if __name__ == "__main__":
    main()
