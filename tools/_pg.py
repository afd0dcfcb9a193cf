import os, sys
sys.path.insert(0, "/root/repo")
import torch
from aki_amd import ops, _lib
lib = _lib.load()
dev = "cuda"
for (M, N, K, act, name) in [(5240, 16384, 3072, ops.ACT_SWIGLU, "gate_up"), (5240, 3072, 8192, 0, "down"), (5240, 3072, 3072, 0, "o_proj"), (5240, 9216, 3072, 0, "qkv plain"), (5240, 32016, 3072, 0, "lm_head"), (4608, 3456, 1152, 0, "siglip qkv")]:
    x = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
    outs = {}
    line = f"{name:12s}"
    best = {0: 1e9, 512: 1e9}
    for rep in range(6):
        for mode in (0, 512):
            lib.aki_debug_set_gemm_tile(mode)
            for _ in range(3): y = ops.linear(x, w, act=act)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): y = ops.linear(x, w, act=act)
            e1.record(); torch.cuda.synchronize()
            outs[mode] = y.clone()
            best[mode] = min(best[mode], e0.elapsed_time(e1) / 20)
    for mode in (0, 512):
        t = best[mode]
        line += f"  mode{mode} {t*1e3:7.1f}us {2.0*M*N*K/t/1e9:6.0f}TF"
    lib.aki_debug_set_gemm_tile(0)
    print(line, " identical:", bool(torch.equal(outs[0], outs[512])))
