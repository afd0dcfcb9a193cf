#!/usr/bin/env python3
"""fp8 configuration, batch 2: `generate` with the step replayed as a graph == issued eagerly, and the e4m3 batched decode step agrees with the
bf16 one within the configuration's tolerance on the first steps' logits."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from aki_amd.factory import build_aki
dev = torch.device("cuda", 0)
m = build_aki(dtype=torch.bfloat16, device=dev, seed=0).eval()
vx, ids, am = bench.synth_batch(2, dev, torch.bfloat16, m.media_token_id, seed=1000)
kw = dict(vision_x=vx, lang_x=ids, attention_mask=am, max_new_tokens=24, do_sample=False, eos_token_id=[])
ref = m.generate(**kw)
m.lang_model.enable_fp8()
a = m.generate(use_graph=True, **kw)
b = m.generate(use_graph=False, **kw)
print("fp8 batch 2: graph == eager:", torch.equal(a, b), "| tokens equal to bf16:", int((a == ref).sum()), "of", a.numel())
