#!/usr/bin/env python3
"""Soak test of the residual-tile prefetch (gemm_bf16.hip: LDS-DMA of the residual tile under the last K-steps, big pipelined tile and
the two-stage small tiles): thousands of launches over shapes with ragged M / N / short K, several in flight, every output compared
BIT FOR BIT with a build of the library that fetches the residual after the K loop (--ref, default aki_amd/lib/abl/libaki_prev.so).
A WAR / RAW slip between the DMA and the fragment reads would show up as sporadic differing tiles.

    python tools/residual_prefetch_soak.py [--seconds 60] [--ref path/to/other/libaki_mi355x.so]
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
    ap.add_argument("--seconds", type=float, default=60.0)
    ap.add_argument("--ref", default=os.path.join(os.path.dirname(_lib.LIB_PATH), "abl", "libaki_prev.so"))
    a = ap.parse_args()
    dev = "cuda"
    new, ref = _lib.load(), _lib._bind(a.ref)
    g = torch.Generator(device=dev).manual_seed(0)
    #        M      N     K      (big pipelined tile: M >= 256 and enough tiles; small tiles otherwise; ragged edges on purpose)
    shapes = [(5240, 3072, 3072), (5240, 3072, 8192), (4608, 1152, 1152), (4608, 1152, 4352), (1380, 3072, 256), (300, 512, 256),
              (2049, 2056, 64), (2049, 2056, 128), (777, 1160, 192), (8 * 655, 1152, 3072), (4096, 4096, 512), (513, 264, 320)]
    data = {}
    for M, N, K in shapes:
        data[(M, N, K)] = (torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16), (torch.randn(N, K, device=dev, generator=g) * 0.05).to(torch.bfloat16),
                           (torch.randn(M, N, device=dev, generator=g) * 2).to(torch.bfloat16), (torch.randn(N, device=dev, generator=g)).to(torch.bfloat16))
    t0, n = time.time(), 0
    while time.time() - t0 < a.seconds:
        for (M, N, K), (x, w, r, b) in data.items():
            for bias, stats in ((None, False), (b, False), (None, True), (b, True)):
                outs = {}
                for name, lib in (("new", new), ("ref", ref)):
                    _lib._lib = lib
                    ys = []
                    for rep in range(3):                      # several launches in flight
                        st = ops.new_stats(M, dev) if stats else None
                        ys.append((ops.linear(x, w, bias=bias, residual=r, stats_out=st, stats_eps=1e-6), st))
                    outs[name] = ys
                _lib._lib = new
                for (y1, s1), (y2, s2) in zip(outs["new"], outs["ref"]):
                    if not torch.equal(y1, y2) or (stats and not torch.equal(s1.rstd, s2.rstd)):
                        bad = int((y1 != y2).sum())
                        raise SystemExit(f"MISMATCH after {n} launches: shape {(M, N, K)} bias={bias is not None} stats={stats}: {bad} elements differ")
                    n += 1
    print(f"{n} residual launches compared bit for bit with {os.path.basename(a.ref)} in {time.time() - t0:.0f} s: OK")


if __name__ == "__main__":
    main()
