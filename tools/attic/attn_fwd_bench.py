"""Time the MMA attention core (forward) at the benchmark shape and at the long-context shape."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from aki_amd import ops
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
for (B, H, L, rects) in [(8, 32, 655, [[(6, 150, 150, 638)]] * 8), (1, 32, 4096, [[(6, 150, 150, 4032), (900, 1044, 1044, 4032), (1800, 1944, 1944, 4032), (2700, 2844, 2844, 4032)]])]:
    q, k, v = (torch.randn(B, H, L, 96, device=dev, generator=g).to(torch.bfloat16) for _ in range(3))
    table = ops.MaskTable.from_host(rects, np.ones((B, L)), [L] * B, dev)
    for _ in range(5):
        ops.mma_attn_core(q, k, v, table, 96 ** -0.5)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        ops.mma_attn_core(q, k, v, table, 96 ** -0.5)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 50
    pairs = L * (L + 1) // 2 + sum(144 * max(0, r[3] - r[2]) for r in rects[0])
    print(f"attn core B{B} L{L}: {ms*1e3:.1f} us  {4.0*H*96*pairs*B/ms/1e9:.0f} TFLOP/s")
