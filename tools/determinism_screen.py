"""Race / hazard screen on the full-size AKI-4B paths: the same inputs through the same weights must give bit-identical
results launch after launch - forward logits (bf16 and fp8), greedy tokens, and the gradients of one training step."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from aki_amd.factory import build_aki

dev = torch.device("cuda", 0)
model = build_aki(dtype=torch.bfloat16, device=dev).eval()
vx, ids, am = bench.synth_batch(8, dev, torch.bfloat16, model.media_token_id, seed=1)
am[1, 400:] = 0
am[5, 300:] = 0
ok = True
with torch.no_grad():
    ref = model(vx, ids, attention_mask=am).logits.clone()
    for i in range(4):
        out = model(vx, ids, attention_mask=am).logits
        nd = int((out != ref).sum())
        ok &= nd == 0
        print(f"forward bf16 launch {i + 1}: {nd} of {out.numel()} logits differ")
    t0 = model.generate(vx[:2], ids[:2], attention_mask=am[:2], max_new_tokens=16)
    for i in range(2):
        t = model.generate(vx[:2], ids[:2], attention_mask=am[:2], max_new_tokens=16)
        same = bool(torch.equal(t, t0))
        ok &= same
        print(f"generate launch {i + 1}: identical tokens {same}")
    model.lang_model.enable_fp8()
    ref8 = model(vx, ids, attention_mask=am).logits.clone()
    for i in range(2):
        out = model(vx, ids, attention_mask=am).logits
        nd = int((out != ref8).sum())
        ok &= nd == 0
        print(f"forward fp8 launch {i + 1}: {nd} logits differ")
    model.lang_model.enable_fp8(False)
from aki_amd.trainer import AkiTrainer
model.train(); model.set_trainable()
tr = AkiTrainer(model)
labels = ids.clone(); labels[labels == model.media_token_id] = -100; labels[am == 0] = -100
grads = []
for i in range(3):
    tr.zero_grad()
    out = model(vx, ids, attention_mask=am, labels=labels)
    tr.backward(out.loss, True)
    torch.cuda.synchronize()
    grads.append((float(out.loss.detach()), tr.g16.clone()))
for i in range(1, 3):
    nd = int((grads[i][1] != grads[0][1]).sum())
    rel = float((grads[i][1].float() - grads[0][1].float()).norm() / grads[0][1].float().norm())
    print(f"training step {i}: loss {grads[i][0]:.6f} vs {grads[0][0]:.6f}; {nd} of {grads[0][1].numel()} gradient elements differ, relative L2 {rel:.3e}")
ok &= all(int((grads[i][1] != grads[0][1]).sum()) == 0 for i in (1, 2))
print("FORWARD / GENERATE / TRAINING STEP DETERMINISTIC" if ok else "NONDETERMINISM FOUND")
# where do the differing gradient elements live?
names = {id(p): n for n, p in model.named_parameters()}
spans = sorted(((lo, hi, names.get(pid, "?")) for pid, (lo, hi) in tr.span_of.items()))
where = {}
for lo, hi, name in spans:
    nd = int((grads[2][1][lo:hi] != grads[0][1][lo:hi]).sum())
    if nd:
        where[name] = nd
print("differing gradient elements by parameter:", where)
