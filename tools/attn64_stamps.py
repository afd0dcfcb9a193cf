"""[optional argument: lab variant, default 612; 613 / 615 / 620 / 621 / 623: with parts taken out - tools/attn64_ablate.py]
Where a workgroup of the 64-row attention core spends its cycles (lab variant 612: s_memtime stamps around the rank prologue, the tile
loops and the epilogue; s_memrealtime over the workgroup gives the clock the chip held).   python tools/attn64_stamps.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from aki_amd import ops, _lib
lab = _lib.load_lab(); _lib._lib = lab
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
IMG4 = [(6, 150, 150, 4032), (900, 1044, 1044, 4032), (1800, 1944, 1944, 4032), (2700, 2844, 2844, 4032)]
for (B, H, L, rects) in [(4, 32, 4096, [IMG4] * 4), (1, 32, 4096, [IMG4]), (4, 32, 4096, [[(0, 0, 0, 0)]] * 4)]:
    q, k, v = (torch.randn(B, H, L, 96, device=dev, generator=g).to(torch.bfloat16) for _ in range(3))
    table = ops.MaskTable.from_host(rects, np.ones((B, L)), None, dev)
    lab.aki_lab_set_attn_variant(int(sys.argv[1]) if len(sys.argv) > 1 else 612)
    for _ in range(30):     # warm: the clock settles under load
        o, lse = ops.mma_attn_core(q, k, v, table, 96 ** -0.5, return_lse=True)
    torch.cuda.synchronize()
    lab.aki_lab_set_attn_variant(0)
    nwg = 256
    d = lse.flatten()[: 8 * nwg].view(nwg, 8).double().cpu().numpy()
    pro, loop, epi, tiles, rt, redo, dma, wait = (d[:, i] for i in range(8))
    tot = pro + loop + epi
    clk = tot / (rt * 10.0)          # cycles per ns = GHz (real-time counter: 100 MHz)
    print(f"B{B} L{L} rects {len(rects[0]) if rects[0][0][1] else 0}: per workgroup (median over {nwg}): total {np.median(tot)/1e3:.0f}k cycles = {np.median(rt)/100:.1f} us at {np.median(clk):.2f} GHz | "
          f"prologue {np.median(pro)/1e3:.1f}k ({100*np.median(pro/tot):.1f} %)  tile loops {np.median(loop)/1e3:.1f}k ({100*np.median(loop/tot):.1f} %)  epilogue {np.median(epi)/1e3:.1f}k ({100*np.median(epi/tot):.1f} %) | "
          f"waiting at the tile barrier {np.median(wait/tiles):.0f} + for the own DMA pieces {np.median(dma/tiles):.0f} cycles per tile (wave 0; both include a stamp's own ~40) | "
          f"{np.median(tiles):.0f} tiles, {np.median(redo):.0f} redos (wave 0), {np.median(loop/tiles):.0f} cycles per tile (56 MFMAs = 1664 pipe cycles) | slowest workgroup {rt.max()/100:.1f} us, fastest {rt.min()/100:.1f} us", flush=True)
