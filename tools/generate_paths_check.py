#!/usr/bin/env python3
"""`AKI.generate` on the full-size model through its three decode paths - eager loop, captured steps, one-launch chain - must give the same
tokens (batch 1: chain eager vs per-layer graph; batch 2: per-layer eager vs graph).   python tools/generate_paths_check.py [--new 40]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from aki_amd.factory import build_aki

ap = argparse.ArgumentParser()
ap.add_argument("--new", type=int, default=40)
a = ap.parse_args()
dev = torch.device("cuda", 0)
m = build_aki(dtype=torch.bfloat16, device=dev, seed=0).eval()
ok = True
for B in (1, 2):
    vx, ids, am = bench.synth_batch(B, dev, torch.bfloat16, m.media_token_id, seed=7)
    outs = {ug: m.generate(vx, ids, attention_mask=am, max_new_tokens=a.new, do_sample=False, eos_token_id=[], use_graph=ug) for ug in (False, True)}
    same = bool(torch.equal(outs[False], outs[True]))
    ok &= same
    print(f"batch {B}: use_graph=True == use_graph=False: {same} {tuple(outs[True].shape)}")
    if B == 1:
        m.lang_model.model.use_decode_chain = False
        o5 = m.generate(vx, ids, attention_mask=am, max_new_tokens=a.new, do_sample=False, eos_token_id=[], use_graph=True)
        m.lang_model.model.use_decode_chain = True
        same = bool(torch.equal(o5, outs[True]))
        ok &= same
        print(f"batch 1: chain == captured per-layer steps: {same}")
sys.exit(0 if ok else 1)
