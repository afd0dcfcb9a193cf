"""Lab: per-phase cycle stamps of the attention core (needs the AKI_ATTN timing build passed as argv[1])."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from aki_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
from aki_amd import ops
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
H, L = 32, 655
for B in (1, 8):
    rects = [[(6, 150, 150, 638)]] * B
    q, k, v = (torch.randn(B, H, L, 96, device=dev, generator=g).to(torch.bfloat16) for _ in range(3))
    table = ops.MaskTable.from_host(rects, np.ones((B, L)), [L] * B, dev)
    for _ in range(3):
        o, lse = ops.mma_attn_core(q, k, v, table, 96 ** -0.5, return_lse=True)
    torch.cuda.synchronize()
    d = lse.view(torch.int32).flatten()[: 24 * 16 * 8].cpu().numpy().astype(np.int64).reshape(6, 4, 16, 8) & 0xFFFFFFFF
    print(f"== B={B} (s_memtime ticks = shader-clock cycles (48k ticks over the ~21 us launch: ~2.3 GHz))")
    for gi in range(6):
        for w in range(4):
            r = d[gi, w]
            t_entry, t_loop, wq0, jend, t_end, t_s1, t_q, t_fin = r[15][:8]
            its = []
            for j in range(min(int(jend), 15)):
                ta0, ta, tb, tc, td, te, kind = r[j][:7]
                its.append(f"{'SFPR'[int(kind)]}:{(ta-ta0)&0xffffffff}/{(tb-ta)&0xffffffff}/{(tc-tb)&0xffffffff}/{(td-tc)&0xffffffff}/{(te-td)&0xffffffff}")
            print(f"g{gi} w{w} wq0={wq0:4d} jend={jend:2d} pro={(t_s1-t_entry)&0xffffffff}/{(t_q-t_s1)&0xffffffff}/{(t_loop-t_q)&0xffffffff} loop={(t_end-t_loop)&0xffffffff:5d} epi={(t_fin-t_end)&0xffffffff:4d} | " + " ".join(its))
