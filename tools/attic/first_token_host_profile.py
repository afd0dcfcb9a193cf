#!/usr/bin/env python3
"""Where the HOST time of one first token goes (AKI.generate, max_new_tokens = 1, batch 1): cProfile of one warm call + wall-clock of the
stage boundaries.  The GPU runs dry twice in a one-sample prefill (profiles/r05_first_token_trace.txt: two gaps of ~0.36 ms in front of the
SigLIP stack and the decoder stack) - this says what Python does there."""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from aki_amd.factory import build_aki
from aki_amd import ops, siglip, phi3, helpers

dev = torch.device("cuda", 0)
model = build_aki(dtype=torch.bfloat16, device=dev, seed=0).eval()
vx, ids, am = bench.synth_batch(1, dev, torch.bfloat16, model.media_token_id, seed=1000)
kw = dict(vision_x=vx, lang_x=ids, attention_mask=am, max_new_tokens=1, do_sample=False)
for _ in range(3):
    model.generate(**kw)
torch.cuda.synchronize()

marks = []
def wrap(mod, name, label):
    f = getattr(mod, name)
    def g(*a, **k):
        t0 = time.perf_counter()
        r = f(*a, **k)
        marks.append((label, t0, time.perf_counter()))
        return r
    setattr(mod, name, g)
wrap(ops, "patch_embed", "patch_embed")
wrap(ops, "params_signature", "params_signature")
wrap(ops, "siglip_stack", "C siglip_stack")
wrap(ops, "perceiver_stack", "C perceiver_stack")
wrap(ops, "decoder_stack", "C decoder_stack")
wrap(ops, "splice", "splice") if hasattr(ops, "splice") else None
for r in range(2):
    marks.clear()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    model.generate(**kw)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"call: host {1e3 * (t1 - t0):.3f} ms, with sync {1e3 * (t2 - t0):.3f} ms")
    for lab, a, b in marks:
        print(f"   {lab:22s} starts {1e6 * (a - t0):8.1f} us, takes {1e6 * (b - a):8.1f} us")
pr = cProfile.Profile()
torch.cuda.synchronize()
pr.enable()
model.generate(**kw)
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print(s.getvalue()[:6000])
