"""Run only the MMA attention core a few times (for rocprofv3 --pmc passes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from aki_amd import ops
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
CASES = [(8, 32, 655, [[(6, 150, 150, 638)]] * 8), (1, 32, 4096, [[(6, 150, 150, 4032), (900, 1044, 1044, 4032)]])]
if len(sys.argv) > 1:          # one case only (both launch the same grid size: PMC averages per grid would mix them)
    CASES = [CASES[int(sys.argv[1])]]
for (B, H, L, rects) in CASES:
    q, k, v = (torch.randn(B, H, L, 96, device=dev, generator=g).to(torch.bfloat16) for _ in range(3))
    table = ops.MaskTable.from_host(rects, np.ones((B, L)), None, dev)
    for _ in range(6):
        ops.mma_attn_core(q, k, v, table, 96 ** -0.5, dead_rows=0)
torch.cuda.synchronize()
