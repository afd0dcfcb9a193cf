#!/usr/bin/env python3
"""The benchmark's forward step under two settings of the lab library's GEMM switch, alternating in one process on one box (boxes differ by up to 5 %).
    python tools/step_ab_lab.py [modeA modeB]      default 0 (product choices) against 2048 (two-stage loops everywhere: no three-deep operand rings)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aki_amd import _lib
lib = _lib.load_lab(); _lib._lib = lib
import bench
modes = [int(a) for a in sys.argv[1:3]] or [0, 2048]
dev = torch.device("cuda:0")
torch.manual_seed(0)
from aki_amd.factory import build_aki
model = build_aki(dtype=torch.bfloat16, device=dev, seed=0)
model.eval()
vx, ids, am = bench.synth_batch(bench.BATCH, dev, torch.bfloat16, model.media_token_id, seed=1000)
def step():
    with torch.no_grad():
        return model(vx, ids, attention_mask=am)
for _ in range(3): step()
torch.cuda.synchronize()
res = {m: [] for m in modes}
for rep in range(6):
    for m in modes:
        lib.aki_lab_set_gemm_tile(m)
        step(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5): step()
        b.record(); torch.cuda.synchronize()
        res[m].append(a.elapsed_time(b) / 5)
lib.aki_lab_set_gemm_tile(0)
for m in modes:
    v = sorted(res[m]); print(f"lab mode {m}: median {v[len(v)//2]:.3f} ms/step, min {v[0]:.3f}")
