"""Does a GEMM of the decoder run slower on weights that come from HBM (as in the real forward: every layer has its own
101 MB gate_up matrix, 226 MB per layer, so nothing survives in the 256 MB Infinity Cache from one step to the next) than
back to back on ONE weight tensor (which stays cache-resident)?  Interleaved A/B in one process, random data.
    python tools/cold_weights_bench.py [n_weights]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aki_amd import ops

dev = "cuda"
NW = int(sys.argv[1]) if len(sys.argv) > 1 else 12
M = 8 * 655
CASES = [("gate_up+swiglu", 16384, 3072, 3), ("qkv-shaped", 9216, 3072, 0), ("down", 3072, 8192, 0), ("o_proj", 3072, 3072, 0)]


def run(x, ws, act, y, iters):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(iters):
        ops.linear(x, ws[i % len(ws)], act=act, out=y)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


g = torch.Generator(device=dev).manual_seed(0)
for name, N, K, act in CASES:
    x = torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16)
    ws = [(torch.randn(N, K, device=dev, generator=g) * 0.02).to(torch.bfloat16) for _ in range(NW)]
    y = torch.empty(M, N // 2 if act == 3 else N, device=dev, dtype=torch.bfloat16)
    for _ in range(3):
        run(x, ws, act, y, NW)
    hot, cold = [], []
    for r in range(5):
        hot.append(run(x, ws[:1], act, y, 2 * NW))
        cold.append(run(x, ws, act, y, 2 * NW))
    fl = 2.0 * M * N * K
    h, c = min(hot), min(cold)
    print(f"{name:16s} N{N} K{K}: one weight {h:7.1f} us {fl/h/1e6:6.0f} TF/s | {NW} rotating weights ({NW*N*K*2/2**20:.0f} MiB) {c:7.1f} us {fl/c/1e6:6.0f} TF/s"
          f" | cold/hot {c/h:.3f}", flush=True)
