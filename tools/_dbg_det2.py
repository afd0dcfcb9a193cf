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
o32 = ops.mma_attn_core(q.float(), k.float(), v.float(), table, 96 ** -0.5).view(B, L, H, 96)
outs = [ops.mma_attn_core(q, k, v, table, 96 ** -0.5).clone().view(B, L, H, 96) for _ in range(6)]
torch.cuda.synchronize()
d = torch.zeros(B, L, H, dtype=torch.bool, device=DEV)
for o in outs[1:]:
    d |= (o != outs[0]).any(-1)
print("rows(b,l,h) ever differing:", int(d.sum()), "of", d.numel())
per_block = d.view(B, -1).sum(0) if False else None
rows = d.any(0).any(-1)     # over b, h -> [L]
blk = torch.arange(L, device=DEV) // 32
for bi in range(21):
    m = blk == bi
    print(f"block {bi:2d} rows {bi*32:3d}..: differing (b,h,row) count {int(d[:, m].sum()):6d}")
# size of differences relative to f32 reference error
e = [(o.float() - o32).abs().max().item() for o in outs]
print("max err vs f32 per run:", [round(x, 5) for x in e])
# lanes? within a 32-row block which row offsets
offs = d.sum((0, 2)).view(-1)[:640].view(20, 32).sum(0)
print("by row offset in block:", offs.tolist())
print("by head:", d.sum((0, 1)).tolist())
print("by batch:", d.sum((1, 2)).tolist())
