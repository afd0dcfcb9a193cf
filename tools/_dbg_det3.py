import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from aki_amd import _lib
if len(sys.argv) > 1:
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
from aki_amd import ops
DEV = "cuda"
B, H, L = 8, 32, 655
g = torch.Generator(device=DEV).manual_seed(7)
q, k, v = (torch.randn(B, H, L, 96, device=DEV, generator=g).to(torch.bfloat16) for _ in range(3))
rects = [[(10, 154, 154, L - 8)]] * B
table = ops.MaskTable.from_host(rects, np.ones((B, L)), [L] * B, DEV)
runs = []
for _ in range(6):
    o, lse = ops.mma_attn_core(q, k, v, table, 96 ** -0.5, return_lse=True)
    runs.append((o.clone().view(B, L, H, 96), lse.clone()))     # lse [B,H,L]
torch.cuda.synchronize()
o0, l0 = runs[0]
for i, (o, l) in enumerate(runs[1:], 1):
    do = (o != o0).any(-1)               # [B,L,H]
    dl = (l != l0).permute(0, 2, 1)      # [B,L,H]
    print(f"run {i}: rows with o diff {int(do.sum())}, lse diff {int(dl.sum())}, both {int((do & dl).sum())}, o-only {int((do & ~dl).sum())}")
    if int(dl.sum()):
        idx = dl.nonzero()[:5]
        for (b, r, h) in idx.tolist():
            print("   b,row,h", b, r, h, "lse", l0[b, h, r].item(), l[b, h, r].item())
    if int((do & ~dl).sum()):
        idx = (do & ~dl).nonzero()[:5]
        for (b, r, h) in idx.tolist():
            dd = (o[b, r, h].float() - o0[b, r, h].float())
            print("   o-only b,row,h", b, r, h, "n elems", int((dd != 0).sum()), "max", dd.abs().max().item(), "which d:", (dd != 0).nonzero().flatten().tolist()[:12])
