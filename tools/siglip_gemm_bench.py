"""GEMM shapes of the SigLIP tower / Perceiver at the benchmark batch under each tile configuration (0 = cost model)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aki_amd import ops, _lib
lib = _lib.load_lab()      # lab twin of the library: same kernels + aki_lab_set_gemm_tile
_lib._lib = lib            # route ops.* through it for this script
dev = "cuda"
shapes = [(4608, 3456, 1152, "siglip qkv"), (4608, 1152, 1152, "siglip out"), (4608, 4304, 1152, "siglip fc1"), (4608, 1152, 4352, "siglip fc2"),
          (5240, 3072, 3072, "lm o_proj"), (5240, 3072, 8192, "lm down"), (5240, 32016, 3072, "lm head"), (1152, 4608, 1152, "perc ff1"), (5760, 1024, 1152, "perc kv"),
          (120, 8192, 3072, "M tail"), (1152, 512, 1152, "perc q"), (1152, 1152, 512, "perc out"), (655, 3072, 3072, "B=1 o_proj")]
for M, N, K, name in shapes:
    x = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
    line = f"{name:12s} M{M} N{N} K{K}: "
    for mode in (1, 2, 3, 0, 256):   # 256 = heuristic without the 4-stage ring
        lib.aki_lab_set_gemm_tile(mode)
        for _ in range(3):
            ops.linear(x, w)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.linear(x, w)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        line += f"  mode{mode} {ms*1e3:7.1f}us {2.0*M*N*K/ms/1e9:6.0f}TF"
    lib.aki_lab_set_gemm_tile(0)
    print(line)
