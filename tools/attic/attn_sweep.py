"""Lab: attention-core time vs batch at the benchmark length (wave quantisation / per-workgroup lifetime)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from aki_amd import _lib
if len(sys.argv) > 1:
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
from aki_amd import ops
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
H, L = 32, 655
for B in (1, 2, 3, 4, 6, 8, 12, 16, 32):
    rects = [[(6, 150, 150, 638)]] * B
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
    pairs = L * (L + 1) // 2 + 144 * 488
    print(f"B{B:3d} WGs {B*H*6:5d}: {ms*1e3:7.1f} us {4.0*H*96*pairs*B/ms/1e9:5.0f} TF/s  per-B {ms*1e3/B:.1f} us")
