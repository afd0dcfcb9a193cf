#!/usr/bin/env python3
"""The pre-training step (forward + backward + clip + AdamW, batch 8) with the weight gradients on aki_gemm_tn (operands as they lie) or through two
transposes + the forward GEMM, alternating on one box.    python tools/train_step_ab_tn.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from aki_amd import train_ops as T
from aki_amd.factory import build_aki
from aki_amd.trainer import AkiTrainer
from aki_amd.phi3 import make_phi3_config
dev = torch.device("cuda:0")
model = build_aki(make_phi3_config(num_hidden_layers=32), dtype=torch.bfloat16, device=dev, seed=0)
model.train(); model.set_trainable()
tr = AkiTrainer(model, lr=1e-4, betas=(0.9, 0.999), weight_decay=0.01, max_grad_norm=1.0)
vx, ids, am = bench.synth_batch(8, dev, torch.bfloat16, model.media_token_id, seed=1000)
labels = ids.clone(); labels[labels == model.media_token_id] = -100
losses = {}
def step(tag=None):
    tr.zero_grad()
    out = model(vx, ids, attention_mask=am, labels=labels)
    tr.backward(out.loss)
    tr.optimizer_step()
    if tag is not None: losses.setdefault(tag, []).append(float(out.loss))
for _ in range(2): step()
torch.cuda.synchronize()
res = {True: [], False: []}
for rep in range(4):
    for m in (True, False):
        T.USE_GEMM_TN = m
        step(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(2): step()
        b.record(); torch.cuda.synchronize()
        res[m].append(a.elapsed_time(b) / 2)
T.USE_GEMM_TN = True
for m in (True, False):
    v = sorted(res[m]); print(f"{'aki_gemm_tn' if m else 'transposes + forward GEMM':28s}: median {v[len(v)//2]:.2f} ms/step, min {v[0]:.2f}   peak memory so far {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
