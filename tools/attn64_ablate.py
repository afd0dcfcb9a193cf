"""Where the 64-row attention core's time goes: timing ablations (lab library, variants 100 + mask; results of ablated builds are wrong by
construction).  Every ablation also drops the redo check (bit 16), so that garbage sums cannot send tiles through the exact path; the
baseline is therefore variant 116.  Bits: 1 no exp / sum / pack, 2 no LDS-DMA, 4 no fragment reads, 8 no tile barrier, 32 every tile fast.
Each variant is warmed and timed on its own (interleaving variants of very different power draw moved the clock under the next one).
READ WITH CARE: a build whose softmax is skipped or whose redo is off produces inf / NaN from the first tile on; MFMAs on such operands
draw less power and the chip holds a higher clock (MI355X guide, DVFS give-back), so such builds look ~15 % faster than their
instruction streams are.  Round 6 chased a "200-cycle branch" for an hour that was this effect; tools/attn64_stamps.py (right data,
in-kernel cycle stamps and the real-time counter) is the instrument to trust.
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
VARS = [1, 9, 10, 164, 116, 117, 124, 148, 131]
NAMES = {164: "every tile through redo (right results)", 10: "64-row exact (right results)", 1: "32-row", 9: "64-row", 116: "no redo check (garbage)", 117: "-softmax", 118: "-dma", 120: "-frag reads", 124: "-barrier", 148: "all fast", 130: "-dma-frag-barrier", 131: "MFMA only"}
for (B, H, L, rects) in [(4, 32, 4096, [IMG4] * 4), (1, 32, 4096, [IMG4])]:
    q, k, v = (torch.randn(B, H, L, 96, device=dev, generator=g).to(torch.bfloat16) for _ in range(3))
    table = ops.MaskTable.from_host(rects, np.ones((B, L)), None, dev)
    out = []
    for var in VARS:
        lab.aki_lab_set_attn_variant(var)
        for _ in range(20):
            ops.mma_attn_core(q, k, v, table, 96 ** -0.5)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            ops.mma_attn_core(q, k, v, table, 96 ** -0.5)
        b.record()
        torch.cuda.synchronize()
        out.append(f"{NAMES[var]} {a.elapsed_time(b) / 20 * 1e3:.1f}")
    lab.aki_lab_set_attn_variant(0)
    print(f"B{B} L{L} rects {len(rects[0]) if rects[0][0][1] else 0}: " + "  ".join(out), flush=True)
