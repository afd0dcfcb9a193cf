"""Where the 64-row attention core's time goes: timing ablations (lab library, variants 100 + mask; results of ablated builds are wrong by
construction).  mask bits: 1 no exp / sum / pack, 2 no LDS-DMA, 4 no fragment reloads, 8 no tile barrier, 16 no row maximum, 32 every tile FULL.
    python tools/attn64_ablate.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from aki_amd import ops, _lib
lab = _lib.load_lab()
_lib._lib = lab
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
IMG4 = [(6, 150, 150, 4032), (900, 1044, 1044, 4032), (1800, 1944, 1944, 4032), (2700, 2844, 2844, 4032)]
VARS = [1, 9, 116, 102, 104, 108, 132, 114]
NAMES = {1: "32-row", 9: "64-row", 101: "-exp", 116: "-max", 117: "-softmax", 102: "-dma", 104: "-frag reads", 108: "-barrier", 132: "all FULL", 114: "-dma-frag-barrier", 163: "MFMA only"}
for (B, H, L, rects) in [(4, 32, 4096, [IMG4] * 4), (4, 32, 4096, [[(0, 0, 0, 0)]] * 4), (4, 32, 4096, [[(6, 150, 150, 4032)]] * 4), (1, 32, 4096, [IMG4]), (1, 32, 4096, [[(0, 0, 0, 0)]])]:
    q, k, v = (torch.randn(B, H, L, 96, device=dev, generator=g).to(torch.bfloat16) for _ in range(3))
    table = ops.MaskTable.from_host(rects, np.ones((B, L)), None, dev)
    best = {v_: 1e9 for v_ in VARS}
    for r in range(4):
        for var in VARS:
            lab.aki_lab_set_attn_variant(var)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(10):
                ops.mma_attn_core(q, k, v, table, 96 ** -0.5)
            b.record()
            torch.cuda.synchronize()
            best[var] = min(best[var], a.elapsed_time(b) / 10 * 1e3)
    lab.aki_lab_set_attn_variant(0)
    print(f"B{B} L{L} rects {len(rects[0]) if rects[0][0][1] else 0}: " + "  ".join(f"{NAMES[v_]} {best[v_]:.1f}" for v_ in VARS), flush=True)
