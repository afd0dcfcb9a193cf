"""Full-size AKI-4B paths on recycled memory full of 0xFF bytes (NaN as bf16/f32): forward (bf16, fp8), generate, one training
step.  Anything that reads unwritten memory, even with weight 0, turns non-finite here."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from aki_amd.factory import build_aki


def poison():
    junk = [torch.full((n,), 0xFF, dtype=torch.uint8, device="cuda") for n in (1 << 16, 1 << 20, 1 << 24, 1 << 28, 1 << 30, 3 << 30, 6 << 30)]
    del junk


dev = torch.device("cuda", 0)
model = build_aki(dtype=torch.bfloat16, device=dev).eval()
vx, ids, am = bench.synth_batch(8, dev, torch.bfloat16, model.media_token_id, seed=1)
am[1, 400:] = 0
am[5, 300:] = 0
with torch.no_grad():
    poison()
    out = model(vx, ids, attention_mask=am)
    print("forward bf16 finite:", bool(torch.isfinite(out.logits.float()).all()))
    poison()
    toks = model.generate(vx[:2], ids[:2], attention_mask=am[:2], max_new_tokens=12)
    print("generate ok:", tuple(toks.shape), int(toks.min()), int(toks.max()))
    poison()
    toks1 = model.generate(vx[:1], ids[:1], attention_mask=am[:1], max_new_tokens=12)
    model.lang_model.enable_fp8()
    poison()
    out8 = model(vx, ids, attention_mask=am)
    print("forward fp8 finite:", bool(torch.isfinite(out8.logits.float()).all()))
    poison()
    toks8 = model.generate(vx[:1], ids[:1], attention_mask=am[:1], max_new_tokens=12)
    print("generate fp8 ok:", tuple(toks8.shape), "agree with bf16 on", int((toks8 == toks1).sum()), "of 12")
    model.lang_model.enable_fp8(False)
from aki_amd.trainer import AkiTrainer
model.train(); model.set_trainable()
tr = AkiTrainer(model)
labels = ids.clone(); labels[labels == model.media_token_id] = -100; labels[am == 0] = -100
poison()
l1 = float(tr.train_step(vx, ids, attention_mask=am, labels=labels))
poison()
l2 = float(tr.train_step(vx, ids, attention_mask=am, labels=labels))
print("train losses:", l1, l2, "grad norm", float(tr.grad_norm()), "weights finite:", bool(torch.isfinite(tr.master).all()))
