"""A/B of attention-core variants in one process (lab library: aki_lab_set_attn_variant 1 = the product kernel - 32-row blocks,
two waves per SIMD; 3 = the same kernel with the software-pipelined tile loop; 2 = 64-row kernel, one
wave per SIMD) at the benchmark shape and the long-context shapes; interleaved rounds, random data, plus the largest output
difference between the two.    python tools/attn_ab.py [variant_a variant_b]   (default 1 3)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from aki_amd import ops, _lib
VA, VB = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1, 3)
lab = _lib.load_lab()
_lib._lib = lab
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
CASES = [(8, 32, 655, [[(6, 150, 150, 638)]] * 8), (16, 32, 655, [[(6, 150, 150, 638)]] * 16), (1, 32, 655, [[(6, 150, 150, 638)]]),
         (1, 32, 4096, [[(6, 150, 150, 4032), (900, 1044, 1044, 4032), (1800, 1944, 1944, 4032), (2700, 2844, 2844, 4032)]]),
         (4, 32, 4096, [[(6, 150, 150, 4032), (900, 1044, 1044, 4032)]] * 4), (8, 32, 207, [[(6, 150, 150, 190)]] * 8)]


def run(q, k, v, table, n):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        o = ops.mma_attn_core(q, k, v, table, 96 ** -0.5)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3, o


for (B, H, L, rects) in CASES:
    q, k, v = (torch.randn(B, H, L, 96, device=dev, generator=g).to(torch.bfloat16) for _ in range(3))
    table = ops.MaskTable.from_host(rects, np.ones((B, L)), None, dev)
    best, outs = {VA: 1e9, VB: 1e9}, {}
    for r in range(5):
        for var in (VA, VB):
            lab.aki_lab_set_attn_variant(var)
            t, o = run(q, k, v, table, 20)
            best[var] = min(best[var], t)
            outs[var] = o
    lab.aki_lab_set_attn_variant(0)
    # visible pairs: causal + rectangles (algorithmic flops = 4 * 96 * pairs per head)
    pairs = L * (L + 1) // 2
    for (r0, r1, c0, c1) in rects[0]:
        for r in range(r0, r1):
            pairs += max(0, c1 - max(c0, r + 1))
    fl = 4.0 * 96 * pairs * B * H
    d = (outs[VA].float() - outs[VB].float()).abs().max().item()
    print(f"B{B} H{H} L{L}: variant {VA} {best[VA]:7.1f} us ({fl/best[VA]/1e6:5.0f} TF/s)   variant {VB} {best[VB]:7.1f} us ({fl/best[VB]/1e6:5.0f} TF/s)   "
          f"{VB}/{VA} time ratio {best[VB]/best[VA]:.3f}   max |difference| {d:.4f}", flush=True)
