"""Time the attention backward (delta + dK/dV + dQ kernels) at the benchmark shape: python tools/attn_bwd_bench.py [L]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from aki_amd import ops, train_ops as T
L = int(sys.argv[1]) if len(sys.argv) > 1 else 655
B, H, Dh = 8, 32, 96
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
q, k, v = (torch.randn(B, H, L, Dh, device=dev, generator=g).to(torch.bfloat16) for _ in range(3))
d_o = torch.randn(B, L, H * Dh, device=dev, generator=g).to(torch.bfloat16)
am = np.ones((B, L), dtype=bool)
table = ops.MaskTable.from_host([[(6, 150, 150, L - 17)]] * B, am, [L] * B, dev)
o, lse = ops.mma_attn_core(q, k, v, table, Dh ** -0.5, return_lse=True)
for _ in range(3):
    T.attn_bwd(q, k, v, o, d_o, lse, table, Dh ** -0.5)
torch.cuda.synchronize()
n = 20
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(n):
    T.attn_bwd(q, k, v, o, d_o, lse, table, Dh ** -0.5)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / n
pairs = L * (L + 1) // 2 + 144 * max(0, (L - 17) - 150)
fl = 2.5 * 4.0 * H * Dh * pairs * B          # backward = 5 products vs the forward's 2
print(f"attn_bwd L={L}: {ms*1e3:.1f} us per call  ({fl/ms/1e9:.0f} TFLOP/s algorithmic)")
