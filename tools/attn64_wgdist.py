import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from aki_amd import ops, _lib
lab = _lib.load_lab(); _lib._lib = lab
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
IMG4 = [(6, 150, 150, 4032), (900, 1044, 1044, 4032), (1800, 1944, 1944, 4032), (2700, 2844, 2844, 4032)]
for (B, rects) in [(4, [IMG4] * 4), (4, [[(0, 0, 0, 0)]] * 4), (1, [IMG4])]:
    H, L = 32, 4096
    q, k, v = (torch.randn(B, H, L, 96, device=dev, generator=g).to(torch.bfloat16) for _ in range(3))
    table = ops.MaskTable.from_host(rects, np.ones((B, L)), None, dev)
    lab.aki_lab_set_attn_variant(612)
    for _ in range(30):
        o, lse = ops.mma_attn_core(q, k, v, table, 96 ** -0.5, return_lse=True)
    torch.cuda.synchronize()
    lab.aki_lab_set_attn_variant(0)
    d = lse.flatten()[: 8 * 256].view(256, 8).double().cpu().numpy()
    rt = d[:, 4] / 100.0
    tiles = d[:, 3]
    loop = d[:, 1]
    print(f"B{B} rects {len(rects[0]) if rects[0][0][1] else 0}: rt us min {rt.min():.0f} med {np.median(rt):.0f} max {rt.max():.0f}; tiles min {tiles.min():.0f} med {np.median(tiles):.0f} max {tiles.max():.0f}")
    for x in range(8):
        m = np.arange(256) % 8 == x
        print(f"  xcd {x}: rt mean {rt[m].mean():.0f} max {rt[m].max():.0f}  tiles mean {tiles[m].mean():.0f}  cycles/tile {np.mean(loop[m]/tiles[m]):.0f}")
    order = np.argsort(-rt)[:8]
    print("  slowest:", [(int(i), round(rt[i]), int(tiles[i]), int(loop[i]/tiles[i])) for i in order])
    order = np.argsort(rt)[:4]
    print("  fastest:", [(int(i), round(rt[i]), int(tiles[i]), int(loop[i]/tiles[i])) for i in order])
