#!/usr/bin/env python3
"""What a GEMM tile's lifetime is made of (lab library: workgroup 0 stamps its K-loop begin and end): prologue, cycles per K-step, epilogue,
for the epilogue combinations the model launches, on one tile alone on the chip (nothing contends) and at the benchmark shapes.

    python tools/gemm_epilogue_probe.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from aki_amd import _lib, ops  # noqa: E402


def main():
    dev = "cuda"
    lib = _lib.load_lab()
    _lib._lib = lib
    probe = torch.zeros(32, dtype=torch.int64, device=dev)
    lib.aki_lab_set_clock_probe(probe.data_ptr())
    g = torch.Generator(device=dev).manual_seed(0)
    rnd = lambda *s, sc=1.0: (torch.randn(*s, device=dev, generator=g) * sc).to(torch.bfloat16)
    f32 = lambda *s: torch.randn(*s, device=dev, generator=g)
    for M, N, K in ((256, 256, 3072), (5240, 3072, 3072), (4608, 1152, 1152), (5240, 8192, 3072)):
        x, w, r, b = rnd(M, K), rnd(N, K, sc=0.02), rnd(M, N), rnd(N, sc=0.1)
        w2 = rnd(2 * N, K, sc=0.02)
        rs = f32(M).abs() + 0.5
        mu, cs = f32(M) * 0.1, f32(N)
        st = ops.new_stats(M, dev)
        stl = ops.new_stats(M, dev, ln=True)
        cases = {
            "plain": lambda: ops.linear(x, w),
            "row scale": lambda: ops.linear(x, w, row_scale=rs),
            "bias": lambda: ops.linear(x, w, bias=b),
            "residual": lambda: ops.linear(x, w, residual=r),
            "residual + stats": lambda: ops.linear(x, w, residual=r, stats_out=st, stats_eps=1e-5),
            "bias + residual + stats": lambda: ops.linear(x, w, bias=b, residual=r, stats_out=stl, stats_eps=1e-6),
            "bias + LN fold": lambda: ops.linear(x, w, bias=b, row_scale=rs, row_shift=mu, col_shift=cs),
            "bias + LN fold + gelu": lambda: ops.linear(x, w, bias=b, row_scale=rs, row_shift=mu, col_shift=cs, act=ops.ACT_GELU_TANH),
            "swiglu + row scale": lambda: ops.linear(x, w2, act=ops.ACT_SWIGLU, row_scale=rs),
        }
        print(f"M {M} N {N} K {K}: {((M + 255) // 256) * ((N + 255) // 256)} tiles of 256 x 256")
        for name, fn in cases.items():
            lib.aki_lab_set_gemm_tile(1)
            for _ in range(20):
                fn()
            torch.cuda.synchronize()
            v = probe.tolist()
            mhz = v[0] / max(v[1], 1) * 100
            nk = K // 64
            print(f"   {name:26s} lifetime {v[0]:7d}  prologue {v[18]:5d}  K loop / step {(v[0] - v[18] - v[19]) / nk:6.0f}  epilogue {v[19]:6d} cycles"
                  f"  ({mhz:.0f} MHz: epilogue {v[19] / mhz:.2f} us)")
    # the fused op's first stage: QKV GEMM with RoPE and the head-major scatter in its epilogue (benchmark shape)
    B, Lq, d, H, Dh = 8, 655, 3072, 32, 96
    x3, wq = rnd(B, Lq, d), rnd(3 * H * Dh, d, sc=0.02)
    cos, sin = f32(Lq, Dh), f32(Lq, Dh)
    rs3 = f32(B * Lq).abs() + 0.5
    pos = torch.arange(Lq, device=dev, dtype=torch.int32).expand(B, Lq).contiguous()
    print(f"qkv + rope, M {B * Lq} N {3 * H * Dh} K {d}")
    for name, kw in (("positions implied", {}), ("position_ids", {"position_ids": pos}), ("position_ids + row scale", {"position_ids": pos, "row_scale": rs3})):
        for mode in (0, 1):
            lib.aki_lab_set_gemm_tile(mode)
            for _ in range(20):
                ops.qkv_rope(x3, wq, cos, sin, H, **kw)
            torch.cuda.synchronize()
            v = probe.tolist()
            mhz = v[0] / max(v[1], 1) * 100
            print(f"   {name:26s} tile mode {mode}: lifetime {v[0]:7d}  prologue {v[18]:5d}  K loop / step {(v[0] - v[18] - v[19]) / (d // 64):6.0f}  epilogue {v[19]:6d} cycles"
                  f"  ({mhz:.0f} MHz: epilogue {v[19] / mhz:.2f} us), of which cos/sin staging {v[20]}")
    # the same GEMM inside the fused op, with and without the attention core behind it
    table = ops.mask_to_table(torch.tril(torch.ones(Lq, Lq, device=dev, dtype=torch.int64)).expand(B, 1, Lq, Lq).contiguous())
    lib.aki_lab_set_gemm_tile(0)
    for _ in range(20):
        ops.mma_attn(x3, wq, cos, sin, table, H, row_scale=rs3)
    torch.cuda.synchronize()
    v = probe.tolist()
    mhz = v[0] / max(v[1], 1) * 100
    print(f"   {'fused op, row scale':26s} tile mode 0: lifetime {v[0]:7d}  prologue {v[18]:5d}  K loop / step {(v[0] - v[18] - v[19]) / (d // 64):6.0f}  epilogue {v[19]:6d} cycles"
          f"  ({mhz:.0f} MHz: epilogue {v[19] / mhz:.2f} us), of which cos/sin staging {v[20]}")

    def loop_us(fn, iters=20):
        a_, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        fn(); torch.cuda.synchronize()
        a_.record()
        for _ in range(iters):
            fn()
        b_.record()
        torch.cuda.synchronize()
        return a_.elapsed_time(b_) / iters * 1e3
    lib.aki_lab_set_clock_probe(None)
    t_sep = sorted(loop_us(lambda: ops.qkv_rope(x3, wq, cos, sin, H, row_scale=rs3)) for _ in range(5))[2]
    t_fused = sorted(loop_us(lambda: ops.mma_attn(x3, wq, cos, sin, table, H, row_scale=rs3)) for _ in range(5))[2]
    print(f"   qkv_rope launch (natural order) {t_sep:7.1f} us; fused op (the same GEMM + attention core) {t_fused:7.1f} us")
    lib.aki_lab_set_gemm_tile(0)


if __name__ == "__main__":
    main()
