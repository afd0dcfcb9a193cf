#!/usr/bin/env python3
"""Time the two attention kernels (decoder core at B8·H32·L655 on the benchmark mask, SigLIP 16x72 at 8 x 576) with whatever
library AKI_MI355X_LIB points at - run it twice, alternating libraries, on ONE box to A/B two builds:

    for i in 1 2 3; do AKI_MI355X_LIB=aki_amd/lib/prev_libaki_mi355x.so python tools/attn_lib_ab.py; python tools/attn_lib_ab.py; done
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from aki_amd import ops  # noqa: E402

dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)


def timeit(fn, iters=50, rounds=5):
    for _ in range(5):
        fn()
    ts = []
    for _ in range(rounds):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters):
            fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / iters * 1e3)
    return min(ts), sorted(ts)[len(ts) // 2]


B, H, L = 8, 32, 655
q, k, v = (torch.randn(B, H, L, 96, device=dev, generator=g).to(torch.bfloat16) for _ in range(3))
table = ops.MaskTable.from_host([[(6, 150, 150, 638)]] * B, np.ones((B, L)), None, dev)
core = timeit(lambda: ops.mma_attn_core(q, k, v, table, 96 ** -0.5))
qkv = torch.randn(8, 576, 3, 16, 72, device=dev, generator=g).to(torch.bfloat16)
nc = timeit(lambda: ops.attention(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], 72 ** -0.5))
print(f"{os.environ.get('AKI_MI355X_LIB', 'product'):45s} core min {core[0]:6.1f} med {core[1]:6.1f} us   siglip min {nc[0]:6.1f} med {nc[1]:6.1f} us")
