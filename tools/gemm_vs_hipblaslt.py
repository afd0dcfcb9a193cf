#!/usr/bin/env python3
"""Yardstick (never on the product path): the HIP GEMM of this library against the vendor library (torch.nn.functional.linear
= hipBLASLt on ROCm) on the decoder's GEMM shapes at the benchmark batch, same box, same process, interleaved rounds on
random data (cdna guide rules 24 / 25).  Both sides compute the plain product y = x W^T in bf16 (no epilogue on either side),
plus - our side only - the fused forms the model actually launches (SwiGLU, residual + row statistics, QKV + RoPE).

    python tools/gemm_vs_hipblaslt.py [--rounds 6] [--iters 20] [--m 5240] [--json out.json]
    python tools/gemm_vs_hipblaslt.py --set siglip --cold      the SigLIP tower (M = 4608) and Perceiver shapes, operands rotated through 8 buffers
    python tools/gemm_vs_hipblaslt.py --set prefill --cold     the decoder at the rows of a one-sample prefill (M = 655 and 207)
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from aki_amd import ops  # noqa: E402

dev = "cuda"


def loop_ms(fn, iters):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=6)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--m", type=int, default=8 * 655)
    ap.add_argument("--json", default=None)
    ap.add_argument("--lab-modes", default="", help="comma-separated aki_lab_set_gemm_tile modes to time as extra arms (lab library), e.g. 1,4")
    ap.add_argument("--shapes", default="", help="comma-separated subset of qkv,o_proj,gate_up,down,lm_head")
    ap.add_argument("--set", default="decoder", choices=["decoder", "siglip", "prefill"])
    ap.add_argument("--cold", action="store_true", help="rotate x / w through 8 buffer sets (operands come out of HBM, as in the forward)")
    a = ap.parse_args()
    M = a.m
    g = torch.Generator(device=dev).manual_seed(0)
    rnd = lambda *s, sc=1.0: (torch.randn(*s, device=dev, generator=g) * sc).to(torch.bfloat16)
    shapes = [("qkv", 9216, 3072), ("o_proj", 3072, 3072), ("gate_up", 16384, 3072), ("down", 3072, 8192), ("lm_head", 32064, 3072)]
    if a.shapes:
        shapes = [sh for sh in shapes if sh[0] in a.shapes.split(",")]
    shapes = [(n_, M, N_, K_) for n_, N_, K_ in shapes]
    if a.set == "siglip":      # 8 images x 576 patches through the tower; the Perceiver's projections at 8 x 144 latents / 8 x 873 keys
        shapes = [("siglip qkv", 4608, 3456, 1152), ("siglip out", 4608, 1152, 1152), ("siglip fc1", 4608, 4304, 1152), ("siglip fc2", 4608, 1152, 4352),
                  ("perceiver kv", 6984, 1024, 1152), ("perceiver ff2", 1152, 1152, 4608)]
    elif a.set == "prefill":
        shapes = [(f"{n_} M{m_}", m_, N_, K_) for m_ in (655, 207) for n_, N_, K_ in
                  (("qkv", 9216, 3072), ("o_proj", 3072, 3072), ("gate_up", 16384, 3072), ("down", 3072, 8192))]
    NBUF = 8 if a.cold else 1
    from aki_amd import _lib as L

    lab_lib = L.load_lab() if a.lab_modes else None
    prod_lib = L.load()

    def lab_arm(mode, fn):          # the lab twin is bound ONCE; an arm only switches the library handle and the tile mode
        def run():
            L._lib = lab_lib
            lab_lib.aki_lab_set_gemm_tile(mode)
            fn()
            lab_lib.aki_lab_set_gemm_tile(0)
            L._lib = prod_lib
        return run
    out = []
    for name, M, N, K in shapes:
        xs, ws = [rnd(M, K) for _ in range(NBUF)], [rnd(N, K, sc=0.02) for _ in range(NBUF)]
        x, w = xs[0], ws[0]
        y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        y2 = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        cnt = [0]

        def nxt():
            cnt[0] += 1
            return xs[cnt[0] % NBUF], ws[cnt[0] % NBUF]
        arms = {"hip_plain": lambda: ops.linear(*nxt(), out=y), "hipblaslt": lambda: (lambda xw: torch.mm(xw[0], xw[1].t(), out=y2))(nxt())}
        if name.startswith("gate_up"):
            ys = torch.empty(M, N // 2, device=dev, dtype=torch.bfloat16)
            arms["hip_fused"] = lambda: ops.linear(*nxt(), act=ops.ACT_SWIGLU, out=ys)
        if name.startswith(("o_proj", "down")):
            res = rnd(M, N)
            st = ops.new_stats(M, dev)
            arms["hip_fused"] = lambda: ops.linear(*nxt(), residual=res, stats_out=st, stats_eps=1e-5, out=y)
        for f in arms.values():     # warm-up (and the vendor library's heuristic / tuning pass)
            for _ in range(3):
                f()
        cnt[0] = -1
        arms["hip_plain"]()
        cnt[0] = -1
        arms["hipblaslt"]()
        torch.cuda.synchronize()
        err = (y.float() - y2.float()).abs().max().item() / max(1e-9, y2.float().abs().max().item())
        for mode in [int(m_) for m_ in a.lab_modes.split(",") if m_]:
            arms[f"hip_lab{mode}_plain"] = lab_arm(mode, arms["hip_plain"])
            if "hip_fused" in arms:
                arms[f"hip_lab{mode}_fused"] = lab_arm(mode, arms["hip_fused"])
            cnt[0] = -1
            arms["hip_plain"]()
            ylab = y.clone()
            cnt[0] = -1
            arms[f"hip_lab{mode}_plain"]()
            torch.cuda.synchronize()
            assert torch.equal(ylab, y), f"lab mode {mode} changes the result of {name}"
            for _ in range(2):
                arms[f"hip_lab{mode}_plain"]()
        t = {k: [] for k in arms}
        for _ in range(a.rounds):
            for k, f in arms.items():
                t[k].append(loop_ms(f, a.iters))
        fl = 2.0 * M * N * K
        rec = dict(case=name, M=M, N=N, K=K, max_rel_diff=round(err, 5))
        for k, v in t.items():
            v.sort()
            rec[k] = dict(min_us=round(v[0] * 1e3, 1), med_us=round(v[len(v) // 2] * 1e3, 1), tflops_min=round(fl / v[0] / 1e9, 1), tflops_med=round(fl / v[len(v) // 2] / 1e9, 1))
        rec["hip_over_vendor_med"] = round(rec["hipblaslt"]["med_us"] / rec["hip_plain"]["med_us"], 4)
        out.append(rec)
        print(json.dumps(rec), flush=True)
    if a.json:
        with open(a.json, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
