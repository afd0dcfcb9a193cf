"""Kernel micro-benchmarks on one MI355X: HIP GEMM vs torch (hipBLASLt) at AKI-4B shapes, MMA attention at config 2.
Interleaved rounds in one process, random data (cdna guide rules 24/25).  Prints one line per case."""
import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aki_amd import ops

dev = "cuda"


def timeit(fn, iters=20, warm=3, rounds=3):
    """Back-to-back launches between one event pair (the GPU stays busy; CPU launch overhead is hidden for kernels
    longer than ~30 us).  Returns (median, min) ms per call over `rounds` rounds."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(rounds):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters):
            fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / iters)
    ts.sort()
    return ts[len(ts) // 2], ts[0]


def gemm_cases():
    M = 8 * 655
    return [("qkv", M, 9216, 3072, 0), ("o_proj", M, 3072, 3072, 0), ("gate_up", M, 16384, 3072, 0),
            ("gate_up_swiglu", M, 16384, 3072, 3), ("down", M, 3072, 8192, 0), ("lm_head", M, 32064, 3072, 0),
            ("siglip_fc1", 8 * 729, 4304, 1152, 0), ("perceiver_ff1", 8 * 144, 4608, 1152, 0)]


def main():
    out = []
    g = torch.Generator(device=dev).manual_seed(0)
    for name, M, N, K, act in gemm_cases():
        x = torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16)
        w = (torch.randn(N, K, device=dev, generator=g) * 0.02).to(torch.bfloat16)
        y = torch.empty(M, N // 2 if act == 3 else N, device=dev, dtype=torch.bfloat16)
        f_hip = lambda: ops.linear(x, w, act=act, out=y)
        f_ref = lambda: torch.nn.functional.linear(x, w)
        t_hip, t_hip_min = timeit(f_hip)
        t_ref, t_ref_min = timeit(f_ref)
        fl = 2.0 * M * N * K
        out.append(dict(case=name, M=M, N=N, K=K, hip_ms=round(t_hip, 4), hip_tflops=round(fl / t_hip / 1e9, 1),
                        blaslt_ms=round(t_ref, 4), blaslt_tflops=round(fl / t_ref / 1e9, 1)))
        print(json.dumps(out[-1]), flush=True)
    # fused qkv+rope
    B, L, H = 8, 655, 32
    x = torch.randn(B, L, 3072, device=dev, generator=g).to(torch.bfloat16)
    w = (torch.randn(9216, 3072, device=dev, generator=g) * 0.02).to(torch.bfloat16)
    pos = torch.arange(L, device=dev)
    inv = 1.0 / (10000 ** (torch.arange(0, 96, 2, device=dev).float() / 96))
    fr = pos[:, None].float() * inv[None]
    cos, sin = torch.cat([fr, fr], -1).cos(), torch.cat([fr, fr], -1).sin()
    t1, _ = timeit(lambda: ops.qkv_rope(x, w, cos, sin, H))
    print(json.dumps(dict(case="qkv_rope_fused", ms=round(t1, 4), tflops=round(2.0 * B * L * 9216 * 3072 / t1 / 1e9, 1))), flush=True)
    q, k, v = ops.qkv_rope(x, w, cos, sin, H)
    import numpy as np
    rects = [[(6, 150, 150, 638)]] * B
    table = ops.MaskTable.from_host(rects, np.ones((B, L)), None, dev)
    P = L * (L + 1) // 2 + 144 * (638 - 150)
    fl = 4.0 * 96 * P * B * H
    t2, t2min = timeit(lambda: ops.mma_attn_core(q, k, v, table, 96 ** -0.5, dead_rows=0))
    print(json.dumps(dict(case="mma_attn_core_L655", ms=round(t2, 4), min_ms=round(t2min, 4), visible_pairs=P,
                          tflops=round(fl / t2 / 1e9, 1), hbm_GBs=round(4 * B * L * 3072 * 2 / t2 / 1e6, 1))), flush=True)
    t3, _ = timeit(lambda: ops.mma_attn(x, w, cos, sin, table, H, dead_rows=0))
    print(json.dumps(dict(case="mma_attn_fused_L655", ms=round(t3, 4))), flush=True)
    t4, _ = timeit(lambda: torch.nn.functional.scaled_dot_product_attention(q, k, v, is_causal=True))
    print(json.dumps(dict(case="torch_sdpa_causal_L655", ms=round(t4, 4))), flush=True)
    # long context
    B, L = 1, 4096
    q = torch.randn(B, H, L, 96, device=dev, generator=g).to(torch.bfloat16)
    k = torch.randn(B, H, L, 96, device=dev, generator=g).to(torch.bfloat16)
    v = torch.randn(B, H, L, 96, device=dev, generator=g).to(torch.bfloat16)
    rects = [[(6, 150, 150, 4032), (900, 1044, 1044, 4032), (1800, 1944, 1944, 4032), (2700, 2844, 2844, 4032)]]
    table = ops.MaskTable.from_host(rects, np.ones((B, L)), None, dev)
    P = L * (L + 1) // 2 + sum(144 * (r[3] - r[2]) for r in rects[0])
    t5, _ = timeit(lambda: ops.mma_attn_core(q, k, v, table, 96 ** -0.5, dead_rows=0))
    print(json.dumps(dict(case="mma_attn_core_L4096_4img", ms=round(t5, 4), tflops=round(4.0 * 96 * P * B * H / t5 / 1e9, 1))), flush=True)
    t6, _ = timeit(lambda: torch.nn.functional.scaled_dot_product_attention(q, k, v, is_causal=True))
    print(json.dumps(dict(case="torch_sdpa_causal_L4096", ms=round(t6, 4))), flush=True)
    # norms
    xx = torch.randn(8 * 655, 3072, device=dev, generator=g).to(torch.bfloat16)
    ww = torch.ones(3072, device=dev, dtype=torch.bfloat16)
    t7, _ = timeit(lambda: ops.rmsnorm(xx, ww, 1e-5))
    print(json.dumps(dict(case="rmsnorm_5240x3072", ms=round(t7, 4), GBs=round(2 * xx.numel() * 2 / t7 / 1e6, 1))), flush=True)


if __name__ == "__main__":
    main()
