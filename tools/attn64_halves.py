"""The four halves of the 64-row core's blind iteration in cycles (lab variant 1636 = stamps around each half; 1637 = the same without softmax VALU):
    slot E first half  = 12 P_B V + 4 row-sum MFMAs (448 MFMA cycles) beside 24 V^T reloads and chunks 0-15 of block A's softmax
    slot E second half = 12 K Q_B^T MFMAs (384) beside 12 K reloads and chunks 16-27
    slot O first half  = 12 P_A V + 4 row-sum MFMAs beside the six LDS-DMA pieces and chunks 0-15 of block B's softmax
    slot O second half = 12 K Q_A^T MFMAs beside chunks 16-27
python tools/attn64_halves.py [variant]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from aki_amd import ops, _lib
lab = _lib.load_lab(); _lib._lib = lab
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
IMG4 = [(6, 150, 150, 4032), (900, 1044, 1044, 4032), (1800, 1944, 1944, 4032), (2700, 2844, 2844, 4032)]
for var in ([int(a) for a in sys.argv[1:]] or [1636, 1637]):
    B, H, L = 4, 32, 4096
    q, k, v = (torch.randn(B, H, L, 96, device=dev, generator=g).to(torch.bfloat16) for _ in range(3))
    table = ops.MaskTable.from_host([IMG4] * B, np.ones((B, L)), None, dev)
    lab.aki_lab_set_attn_variant(var)
    for _ in range(30):
        o, lse = ops.mma_attn_core(q, k, v, table, 96 ** -0.5, return_lse=True)
    torch.cuda.synchronize()
    lab.aki_lab_set_attn_variant(0)
    d = lse.flatten()[: 8 * 256].view(256, 8).double().cpu().numpy()
    n = d[:, 3]
    e1, e2, o1, o2 = d[:, 0] / n, d[:, 1] / n, d[:, 2] / n, d[:, 5] / n
    slow = d[:, 6] / np.maximum(d[:, 7], 1)
    print(f"variant {var}: bias / exact iterations per workgroup {np.median(d[:, 7]):.0f} of {np.median(d[:, 7] + n):.0f} (wave 0), {np.median(slow):.0f} cycles each, barrier wait and one stamp included "
          f"= {100 * np.median(d[:, 6] / (d[:, 6] + d[:, 0] + d[:, 1] + d[:, 2] + d[:, 5])):.0f} % of the stamped tile time", flush=True)
    print(f"variant {var}: blind iterations per workgroup {np.median(n):.0f} (wave 0) | cycles per half (median; each includes one stamp, ~40): "
          f"E.1 {np.median(e1):.0f} (MFMA 448)  E.2 {np.median(e2):.0f} (384)  O.1 {np.median(o1):.0f} (448)  O.2 {np.median(o2):.0f} (384) | sum {np.median(e1 + e2 + o1 + o2):.0f} | a whole blind iteration, barrier wait and six stamps included: {np.median(d[:, 4] / n):.0f}", flush=True)
