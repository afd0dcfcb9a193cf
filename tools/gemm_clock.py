#!/usr/bin/env python3
"""Engine clock the big GEMMs actually run at (lab library: workgroup 0 of every bf16 GEMM launch leaves its lifetime in
shader cycles and in 100 MHz wall ticks).  A sustained loop of the decoder's GEMMs, then per shape: launch time from HIP
events, clock = cycles / ticks * 100 MHz, and TFLOP/s against the dense peak AT THAT CLOCK (256 CUs x 4 SIMDs x 1024 FLOP/clk).

    python tools/gemm_clock.py [--warm-seconds 3]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from aki_amd import _lib, ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--warm-seconds", type=float, default=3.0)
    a = ap.parse_args()
    dev = "cuda"
    M, d, F = 8 * 655, 3072, 8192
    g = torch.Generator(device=dev).manual_seed(0)
    rnd = lambda *s, sc=1.0: (torch.randn(*s, device=dev, generator=g) * sc).to(torch.bfloat16)
    h, o, act = rnd(M, d), rnd(M, d), rnd(M, F)
    wo, wg, wd = rnd(d, d, sc=0.02), rnd(2 * F, d, sc=0.02), rnd(d, F, sc=0.02)
    cases = [("o_proj  N3072 K3072", lambda: ops.linear(o, wo, residual=h), 2.0 * M * d * d),
             ("gate_up N16384 K3072 +swiglu", lambda: ops.linear(h, wg, act=ops.ACT_SWIGLU), 2.0 * M * 2 * F * d),
             ("down    N3072 K8192", lambda: ops.linear(act, wd, residual=h), 2.0 * M * d * F)]
    with _lib.use_lab(0) as lib:
        probe = torch.zeros(32, dtype=torch.int64, device=dev)
        lib.aki_lab_set_clock_probe(probe.data_ptr())
        t0 = time.time()
        while time.time() - t0 < a.warm_seconds:          # bring the package to its sustained power state
            for _, fn, _ in cases:
                fn()
            torch.cuda.synchronize()
        for name, fn, fl in cases:
            clocks, times = [], []
            for _ in range(20):
                for _ in range(10):
                    fn()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                fn()
                e1.record()
                torch.cuda.synchronize()
                c, w = probe.tolist()[:2]
                clocks.append(c / max(w, 1) * 100.0)
                times.append(e0.elapsed_time(e1))
            clocks.sort()
            times.sort()
            mhz, ms = clocks[len(clocks) // 2], times[len(times) // 2]
            peak = 256 * 4 * 1024 * mhz * 1e6 / 1e12
            print(f"{name:32s} {ms * 1e3:7.1f} us  {fl / ms / 1e9:7.1f} TF/s   engine clock {mhz:6.0f} MHz   "
                  f"peak at that clock {peak:6.0f} TF/s -> {fl / ms / 1e9 / peak:5.3f}")
        lib.aki_lab_set_clock_probe(None)


if __name__ == "__main__":
    main()
