import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from aki_amd import _lib
if len(sys.argv) > 1:
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
from aki_amd import ops
DEV = "cuda"
for (B, H, L) in [(3, 24, 2560), (8, 32, 655), (2, 4, 300)]:
    g = torch.Generator(device=DEV).manual_seed(7)
    q, k, v = (torch.randn(B, H, L, 96, device=DEV, generator=g).to(torch.bfloat16) for _ in range(3))
    rects = [[(10, 154, 154, L - 8)]] * B
    table = ops.MaskTable.from_host(rects, np.ones((B, L)), [L] * B, DEV)
    outs = [ops.mma_attn_core(q, k, v, table, 96 ** -0.5).clone() for _ in range(4)]
    torch.cuda.synchronize()
    nd = [(outs[0] != o).sum().item() for o in outs[1:]]
    print(sys.argv[1:] , (B, H, L), "elements differing from run 0:", nd)
