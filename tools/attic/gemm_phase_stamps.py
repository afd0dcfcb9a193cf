#!/usr/bin/env python3
"""Where the hand-placed one-wave-per-SIMD GEMM loop (lab tile mode 4) spends its K loop: shader cycles per wave of workgroup 0 in
{top wait + barrier, first half-step (64 MFMAs = 1024 matrix-core cycles), mid wait + barrier, second half-step}, per K-step.

    python tools/gemm_phase_stamps.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from aki_amd import _lib, ops  # noqa: E402


def main():
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    rnd = lambda *s, sc=1.0: (torch.randn(*s, device=dev, generator=g) * sc).to(torch.bfloat16)
    M = 8 * 655
    lib = _lib.load_lab()
    _lib._lib = lib
    probe = torch.zeros(32, dtype=torch.int64, device=dev)
    for name, N, K in (("qkv", 9216, 3072), ("o_proj", 3072, 3072), ("down", 3072, 8192)):
        x, w = rnd(M, K), rnd(N, K, sc=0.02)
        lib.aki_lab_set_gemm_tile(4)
        lib.aki_lab_set_clock_probe(probe.data_ptr())
        for _ in range(30):
            ops.linear(x, w)
        torch.cuda.synchronize()
        v = probe.tolist()
        nk = K // 64
        life, wall = v[0], v[1]
        print(f"{name}: K-steps {nk}, workgroup 0 lifetime {life} cycles ({life / max(wall, 1) * 100:.0f} MHz), per K-step {life / nk:.0f}")
        for wv in range(4):
            a, b, c, d = (v[2 + 4 * wv + i] / nk for i in range(4))
            print(f"   wave {wv}: top wait+barrier {a:6.0f}   half-step 0 {b:6.0f}   mid wait+barrier {c:6.0f}   half-step 1 {d:6.0f}   sum {a + b + c + d:6.0f}  (ideal 2 x 1024)")
        print(f"   prologue (entry -> K loop) {v[18]} cycles, epilogue (K loop end -> exit) {v[19]} cycles")
        for mode in (1, 0):
            lib.aki_lab_set_gemm_tile(mode)
            for _ in range(30):
                ops.linear(x, w)
            torch.cuda.synchronize()
            v = probe.tolist()
            print(f"   tile mode {mode} (8 waves): lifetime {v[0]} cycles, per K-step {v[0] / nk:.0f}; prologue {v[18]}, K loop {v[0] - v[18] - v[19]} "
                  f"({(v[0] - v[18] - v[19]) / nk:.0f} per step), epilogue {v[19]}")
        lib.aki_lab_set_clock_probe(None)
        lib.aki_lab_set_gemm_tile(0)


if __name__ == "__main__":
    main()
