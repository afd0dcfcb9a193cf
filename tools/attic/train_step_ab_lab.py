#!/usr/bin/env python3
"""The pre-training step (forward + backward + clip + AdamW, batch 8) under two settings of the lab library's GEMM switch, alternating on one box.
    python tools/train_step_ab_lab.py [modeA modeB]    default 0 against 2048 (two-stage loops everywhere)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aki_amd import _lib
lib = _lib.load_lab(); _lib._lib = lib
import bench
from aki_amd.factory import build_aki
from aki_amd.trainer import AkiTrainer
from aki_amd.phi3 import make_phi3_config
modes = [int(a) for a in sys.argv[1:3]] or [0, 2048]
dev = torch.device("cuda:0")
model = build_aki(make_phi3_config(num_hidden_layers=32), dtype=torch.bfloat16, device=dev, seed=0)
model.train(); model.set_trainable()
tr = AkiTrainer(model, lr=1e-4, betas=(0.9, 0.999), weight_decay=0.01, max_grad_norm=1.0)
vx, ids, am = bench.synth_batch(8, dev, torch.bfloat16, model.media_token_id, seed=1000)
labels = ids.clone(); labels[labels == model.media_token_id] = -100
def step():
    tr.zero_grad()
    out = model(vx, ids, attention_mask=am, labels=labels)
    tr.backward(out.loss)
    tr.optimizer_step()
for _ in range(2): step()
torch.cuda.synchronize()
res = {m: [] for m in modes}
for rep in range(4):
    for m in modes:
        lib.aki_lab_set_gemm_tile(m)
        step(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(2): step()
        b.record(); torch.cuda.synchronize()
        res[m].append(a.elapsed_time(b) / 2)
lib.aki_lab_set_gemm_tile(0)
for m in modes:
    v = sorted(res[m]); print(f"lab mode {m}: median {v[len(v)//2]:.2f} ms/step, min {v[0]:.2f}")
