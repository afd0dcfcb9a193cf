import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from aki_amd import ops
DEV = "cuda"
B, H, L = 3, 24, 2560
g = torch.Generator(device=DEV).manual_seed(7)
q, k, v = (torch.randn(B, H, L, 96, device=DEV, generator=g).to(torch.bfloat16) for _ in range(3))
rects = [[(10, 154, 154, L - 8)], [(0, 0, 0, 0)], [(300, 444, 444, 2000), (900, 1044, 1044, 2000)]]
am = np.ones((B, L)); am[1, L - 100:] = 0
seqs = [L, L, L - 37]
table = ops.MaskTable.from_host(rects, am, seqs, DEV)
o1 = ops.mma_attn_core(q, k, v, table, 96 ** -0.5)
o2 = ops.mma_attn_core(q, k, v, table, 96 ** -0.5)
print("deterministic:", torch.equal(o1, o2))
o32 = ops.mma_attn_core(q.float(), k.float(), v.float(), table, 96 ** -0.5)
print("max err vs f32:", (o1.float() - o32).abs().max().item())
for b in range(B):
    tb = ops.MaskTable.from_host([rects[b]], am[b:b + 1], [seqs[b]], DEV)
    ob = ops.mma_attn_core(q[b:b + 1].contiguous(), k[b:b + 1].contiguous(), v[b:b + 1].contiguous(), tb, 96 ** -0.5)
    d = (o1[b:b + 1].float() - ob.float()).abs()          # [1, L, H*96]
    bad = (d > 0).nonzero()
    print(b, "n diff", bad.shape[0], "max", d.max().item())
    if bad.shape[0]:
        rows = torch.unique(bad[:, 1]); heads = torch.unique(bad[:, 2] // 96)
        print("  rows", rows[:20].tolist(), "... count", rows.numel(), " heads", heads.tolist()[:24])
        e1 = (o1[b].float() - o32[b]).abs().max().item(); e2 = (ob[0].float() - o32[b]).abs().max().item()
        print("  err batch-run vs f32", e1, " single-run vs f32", e2)
