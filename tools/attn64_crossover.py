"""Where the 64-row attention core overtakes the 32-row kernel: both forced (lab variants 1 / 9) at lengths between 1024 and 2304, one image
per sample, batch 1 / 4 / 8; best of 5 x 10 launches.  The product rule (AKI_ATTN64_MIN_L in mma_attn_bf16.hip) follows this table.
    python tools/attn64_crossover.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from aki_amd import ops, _lib
lab = _lib.load_lab(); _lib._lib = lab
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)


def run(q, k, v, table, n):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        ops.mma_attn_core(q, k, v, table, 96 ** -0.5)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for B in (1, 4, 8):
    row = []
    for L in (1024, 1280, 1536, 1792, 2048, 2304):
        q, k, v = (torch.randn(B, 32, L, 96, device=dev, generator=g).to(torch.bfloat16) for _ in range(3))
        table = ops.MaskTable.from_host([[(6, 150, 150, L - 64)]] * B, np.ones((B, L)), None, dev)
        t = {}
        for var in (1, 9):
            lab.aki_lab_set_attn_variant(var)
            run(q, k, v, table, 5)
            t[var] = min(run(q, k, v, table, 10) for _ in range(5))
        lab.aki_lab_set_attn_variant(0)
        row.append(f"L{L}: {t[1]:.1f} / {t[9]:.1f} us ({t[9] / t[1]:.2f})")
    print(f"B{B} H32, 32-row / 64-row (ratio): " + "  ".join(row), flush=True)
