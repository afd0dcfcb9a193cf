"""The attention core alone, for a profiler: N launches of aki_mma_attn_core_fwd at (B, 32 heads, L) with `n_img` interleaved 144-row images
(BASELINE configs[3]'s mask at L = 4096, n_img = 4), random bf16 data, through the product library.   python tools/attn_core_run.py B L n_img [N]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from aki_amd import ops
B, L, n_img = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
N = int(sys.argv[4]) if len(sys.argv) > 4 else 20
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
starts = [6, 900, 1800, 2700][:n_img]
rects = [[(s, s + 144, s + 144, L - 64) for s in starts] or [(0, 0, 0, 0)]] * B
q, k, v = (torch.randn(B, 32, L, 96, device=dev, generator=g).to(torch.bfloat16) for _ in range(3))
table = ops.MaskTable.from_host(rects, np.ones((B, L)), None, dev)
for _ in range(N):
    ops.mma_attn_core(q, k, v, table, 96 ** -0.5)
torch.cuda.synchronize()
pairs = L * (L + 1) // 2 + sum(144 * max(0, (L - 64) - (s + 144)) for s in starts)
print(f"B{B} L{L} images {n_img}: {4.0 * 96 * pairs * B * 32 / 1e9:.2f} GFLOP and {4 * B * L * 3072 * 2 / 1e6:.1f} MB algorithmic per launch")
