#!/usr/bin/env python3
"""The benchmark's forward, again and again on the same batch: the logits must be bit-identical with the first pass.   python tools/forward_soak.py [--passes 300]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from aki_amd.factory import build_aki
ap = argparse.ArgumentParser(); ap.add_argument("--passes", type=int, default=300); a = ap.parse_args()
dev = torch.device("cuda:0")
model = build_aki(dtype=torch.bfloat16, device=dev, seed=0); model.eval()
vx, ids, am = bench.synth_batch(bench.BATCH, dev, torch.bfloat16, model.media_token_id, seed=1000)
ref = None; bad = 0
with torch.no_grad():
    for i in range(a.passes):
        lg = model(vx, ids, attention_mask=am).logits
        if ref is None: ref = lg.clone()
        elif not torch.equal(lg, ref): bad += 1
print(f"{a.passes} forwards, {bad} differ from the first")
