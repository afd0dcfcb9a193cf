"""Lab: time the attention core from an alternative build of the library (argv[1])."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from aki_amd import _lib
if len(sys.argv) > 1:
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
from aki_amd import ops
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
out = []
for (B, H, L, rects) in [(8, 32, 655, [[(6, 150, 150, 638)]] * 8), (8, 32, 4096, [[]] * 8), (32, 32, 1024, [[]] * 32)]:
    q, k, v = (torch.randn(B, H, L, 96, device=dev, generator=g).to(torch.bfloat16) for _ in range(3))
    table = ops.MaskTable.from_host(rects, np.ones((B, L)), [L] * B, dev)
    for _ in range(3):
        ops.mma_attn_core(q, k, v, table, 96 ** -0.5)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.mma_attn_core(q, k, v, table, 96 ** -0.5)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    pairs = L * (L + 1) // 2 + (sum(144 * max(0, r[3] - r[2]) for r in rects[0]) if rects[0] else 0)
    out.append(f"B{B} L{L}: {ms*1e3:.1f} us {4.0*H*96*pairs*B/ms/1e9:.0f} TF/s")
print(os.path.basename(sys.argv[1]) if len(sys.argv) > 1 else "default", " | ".join(out))
