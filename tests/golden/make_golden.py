#!/usr/bin/env python3
"""Generate the committed golden vectors by running the reference itself (build container only).

    python tests/golden/make_golden.py           # rewrites tests/golden/*.npz

What runs: the first-party reference files under /root/reference (imported through
``ref_import.py``) plus the installed transformers Phi-3 / SigLIP modules the reference delegates to,
with the transformers==4.41.2 4-D mask inversion applied by the harness.
Outputs are small .npz files of inputs that cannot be regenerated from a seed plus expected outputs;
weights are regenerated on both sides by ``gen.fill_params``.
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen  # noqa: E402
import ref_import as R  # noqa: E402

torch.set_grad_enabled(False)
torch.manual_seed(0)


def shapes_of(module):
    return [(k, tuple(v.shape)) for k, v in module.state_dict().items()]


def load_filled(module, seed):
    shapes = shapes_of(module)
    params = gen.fill_params(shapes, seed)
    module.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=True)
    return shapes


def save(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrays)
    print(f"wrote {name}: {os.path.getsize(path) / 1024:.1f} KiB")


def g_mask(ref):
    """a7: VLMWithLanguageStream._make_modality_mutual_mask (src/vlm.py:410-443)."""
    fn = ref.vlm.VLMWithLanguageStream._make_modality_mutual_mask
    cases = gen.mask_cases()
    out = {"n_cases": np.array(len(cases))}
    for i, (am, s, t, e) in enumerate(cases):
        amt = torch.from_numpy(am)
        m = fn(attention_mask_2d=amt, image_start_idx=s, text_start_idx=t, text_end_idx=e,
               input_ids_shape=amt.shape, dtype=amt.dtype, device=amt.device)
        assert m.dtype == torch.int64 and tuple(m.shape) == (1, len(am), len(am))
        out[f"am_{i}"] = am.astype(np.int8)
        out[f"args_{i}"] = np.array([s, t, e], dtype=np.int64)
        out[f"bits_{i}"] = gen.pack_mask_bits(m.numpy())
    save("mask_cases.npz", **out)


def build_tiny(ref, seed=11):
    from transformers import Phi3Config, Phi3ForCausalLM, SiglipVisionConfig
    import transformers.models.siglip.modeling_siglip as ms
    T = gen.TINY
    cfg = Phi3Config(vocab_size=T["vocab"], hidden_size=T["lm_hidden"], intermediate_size=T["lm_inter"],
                     num_hidden_layers=T["lm_layers"], num_attention_heads=T["lm_heads"],
                     num_key_value_heads=T["lm_heads"], max_position_embeddings=4096,
                     original_max_position_embeddings=4096, pad_token_id=T["pad_token_id"],
                     attn_implementation="eager")
    lm = Phi3ForCausalLM(cfg)
    vcfg = SiglipVisionConfig(hidden_size=T["vis_hidden"], intermediate_size=T["vis_inter"],
                              num_hidden_layers=T["vis_layers"], num_attention_heads=T["vis_heads"],
                              image_size=T["image"], patch_size=T["patch"], attn_implementation="eager",
                              vision_use_head=False)
    vt = ms.SiglipVisionTransformer(vcfg)
    model = ref.aki.AKI(vision_encoder=vt, lang_model=lm, vis_feature_dim=T["vis_hidden"],
                        initial_tokenizer_len=T["vocab"], pad_token_id=T["pad_token_id"],
                        decoder_layers_attr_name="model.layers", num_vision_tokens=T["num_vision_tokens"])
    model.lang_model.config.vocab_size = T["vocab"] + 2                      # src/factory.py:144
    model.set_special_token_ids({"<image>": T["media_token_id"], "<|endofchunk|>": T["eoc_token_id"]})
    shapes = load_filled(model, seed)
    R.wrap_lm_441_mask(model.lang_model)
    model.eval()
    return model, shapes


def tiny_batch():
    T = gen.TINY
    rng = gen.rng_for("tiny_batch")
    IMG, A, END, PAD, EOS = T["media_token_id"], 32001, 32007, T["pad_token_id"], 2
    rows = [
        [1, 32006, 5, 6, END, 32010, IMG, 7, 8, 9, 10, 11, END, A, 12, 13, 14, EOS],   # chat prompt, MMA fires
        [1, IMG, 20, 21, 22, 23, T["eoc_token_id"], EOS],                               # pre-training caption: causal
        [1, 30, 31, 32, 33, A, 34, 35, EOS],                                            # no image in this sample
        [1, 32010, IMG, 40, 41, 42, 43, 44, 45, 46, 47, END, A, 48, EOS],
    ]
    n = max(len(r) for r in rows)
    lang_x = np.full((len(rows), n), PAD, dtype=np.int64)
    am = np.zeros((len(rows), n), dtype=np.int64)
    for i, r in enumerate(rows):
        lang_x[i, :len(r)] = r
        am[i, :len(r)] = 1
    labels = lang_x.copy()
    labels[labels == PAD] = -100                                                         # train/losses.py:98-99
    vision_x = rng.standard_normal((len(rows), 1, 1, 3, T["image"], T["image"]), dtype=np.float32)
    return lang_x, am, labels, vision_x


def g_tiny_e2e(ref):
    """a1: AKI.forward end to end + a6 splice outputs + a4/a2 intermediate tensors."""
    model, shapes = build_tiny(ref)
    lang_x, am, labels, vision_x = tiny_batch()
    tl, ta, tlab, tv = map(torch.from_numpy, (lang_x, am, labels, vision_x))
    feats = model._encode_vision_x(tv)
    vtok = model.vision_tokenizer(feats)
    prep = model._prepare_inputs_for_forward(vision_tokens=vtok, lang_x=tl, attention_mask=ta, labels=tlab,
                                             padding_side="right")
    out = model(tv, tl, attention_mask=ta, labels=tlab)
    cols = np.unique(np.concatenate((np.arange(0, 32013, 997), [1, 2, 32000, 32001, 32007, 32010, 32011, 32012])))
    # the reference's own bf16 eager run of the same model (a yardstick for the bf16 HIP path's end-to-end error)
    import copy
    m16 = copy.deepcopy(model).to(torch.bfloat16)
    out16 = m16(tv.to(torch.bfloat16), tl, attention_mask=ta, labels=tlab)
    save("tiny_e2e.npz",
         shapes=np.array(json.dumps(shapes)),
         lang_x=lang_x, attention_mask=am, labels=labels,
         vision_feats=feats.numpy(), vision_tokens=vtok.numpy(),
         inputs_embeds=prep["inputs_embeds"].numpy(),
         mask_bits=gen.pack_mask_bits(prep["attention_mask"].numpy()),
         mask_shape=np.array(prep["attention_mask"].shape),
         new_labels=prep["labels"].numpy(),
         logit_cols=cols, logits=out.logits[:, :, cols].numpy(), loss=np.array(float(out.loss)),
         logits16=out16.logits[:, :, cols].float().numpy(), loss16=np.array(float(out16.loss)))
    # left-padded variant used by generate() (src/aki.py:171)
    prep_l = model._prepare_inputs_for_forward(vision_tokens=vtok, lang_x=tl, attention_mask=ta, padding_side="left")
    save("tiny_splice_left.npz", inputs_embeds=prep_l["inputs_embeds"].numpy(),
         mask_bits=gen.pack_mask_bits(prep_l["attention_mask"].numpy()), mask_shape=np.array(prep_l["attention_mask"].shape))


def g_tiny_grads(ref):
    """a13/a14: gradients of the reference's own loss.backward() on the tiny model (fp32 eager autograd).  Per trainable
    parameter: sum, sum of |.|, and the values at gen.grad_sample_idx(numel) (full tensors would be 25 MB)."""
    model, shapes = build_tiny(ref)
    model.set_trainable()
    lang_x, am, labels, vision_x = tiny_batch()
    tl, ta, tlab, tv = map(torch.from_numpy, (lang_x, am, labels, vision_x))
    model.zero_grad()
    with torch.enable_grad():
        out = model(tv, tl, attention_mask=ta, labels=tlab)
        out.loss.backward()
    names, sums, abss, samples = [], [], [], []
    for n_, p_ in model.named_parameters():
        if not p_.requires_grad:
            continue
        g_ = p_.grad if p_.grad is not None else torch.zeros_like(p_)
        g64 = g_.double().flatten()
        idx = gen.grad_sample_idx(g64.numel())
        smp = np.zeros(48, dtype=np.float32)
        smp[: len(idx)] = g_.detach().flatten()[torch.from_numpy(idx)].numpy()
        names.append(n_); sums.append(float(g64.sum())); abss.append(float(g64.abs().sum())); samples.append(smp)
    save("tiny_grads.npz", names=np.array(json.dumps(names)), sums=np.array(sums), abss=np.array(abss),
         samples=np.stack(samples), loss=np.array(float(out.loss)))


def g_decoupled(ref):
    """a5/a12: DecoupledEmbedding / DecoupledLinear (src/helpers.py:445-484, 594-603)."""
    H = ref.helpers
    rng = gen.rng_for("decoupled")
    V, d, extra = 50, 24, 3
    W = rng.standard_normal((V + 7, d), dtype=np.float32)      # weight has more rows than max_original_id+1
    emb = H.DecoupledEmbedding(max_original_id=V - 1, num_additional_embeddings=extra, _weight=torch.from_numpy(W.copy()),
                               pad_token_id=5)
    Wadd = rng.standard_normal((extra, d), dtype=np.float32)
    emb.additional_embedding.weight.data.copy_(torch.from_numpy(Wadd))
    ids = rng.integers(0, V + extra, size=(3, 17)).astype(np.int64)
    e = emb(torch.from_numpy(ids)).numpy()
    lin = H.DecoupledLinear(max_original_id=V - 1, additional_out_features=extra, _weight=torch.from_numpy(W.copy()),
                            _bias=None, bias=False)
    Wl = rng.standard_normal((extra, d), dtype=np.float32)
    lin.additional_fc.weight.data.copy_(torch.from_numpy(Wl))
    x = rng.standard_normal((2, 5, d), dtype=np.float32)
    y = lin(torch.from_numpy(x)).numpy()
    save("decoupled.npz", W=W, Wadd=Wadd, ids=ids, emb=e, Wl=Wl, x=x, y=y, V=np.array(V))


def g_perceiver(ref):
    """a4: PerceiverResampler small (full output) and at AKI-4B dimensions (row subset)."""
    H = ref.helpers
    m = H.PerceiverResampler(dim=64, dim_inner=192, num_latents=8)
    shapes = load_filled(m, 21)
    x = gen.rng_for("perceiver_small").standard_normal((2, 1, 1, 16, 64), dtype=np.float32)
    y = m(torch.from_numpy(x)).numpy()
    save("perceiver_small.npz", shapes=np.array(json.dumps(shapes)), y=y)
    m = H.PerceiverResampler(dim=1152, dim_inner=3072, num_latents=144)
    shapes = load_filled(m, 22)
    x = gen.rng_for("perceiver_full").standard_normal((1, 1, 1, 729, 1152), dtype=np.float32)
    y = m(torch.from_numpy(x)).numpy()
    rows = np.array([0, 1, 17, 71, 100, 143])
    save("perceiver_full.npz", shapes=np.array(json.dumps(shapes)), rows=rows, y_rows=y[0, 0, rows],
         y_sum=np.array(float(y.astype(np.float64).sum())), y_abs=np.array(float(np.abs(y.astype(np.float64)).sum())))


def g_attn_block(ref):
    """a9/a10: Phi3Attention under the reference's MMA mask (4.41.2 semantics), fp32 and bf16 eager."""
    from transformers import Phi3Config
    from transformers.models.phi3.modeling_phi3 import Phi3Attention, Phi3RotaryEmbedding
    fn = ref.vlm.VLMWithLanguageStream._make_modality_mutual_mask
    for tag, (B, L, H) in {"small": (2, 40, 2), "mid": (2, 200, 4)}.items():
        d = 96 * H
        cfg = Phi3Config(hidden_size=d, num_attention_heads=H, num_key_value_heads=H, intermediate_size=64,
                         num_hidden_layers=1, max_position_embeddings=4096, original_max_position_embeddings=4096,
                         attn_implementation="eager")
        attn = Phi3Attention(cfg, layer_idx=0).eval()
        shapes = load_filled(attn, 31)
        rot = Phi3RotaryEmbedding(cfg)
        rng = gen.rng_for("attn_block_" + tag)
        x = rng.standard_normal((B, L, d), dtype=np.float32)
        # sample 0: image at 3, Nv = L//4, <|assistant|> late, right padding; sample 1: full length, different span
        Nv = L // 4
        am = np.ones((B, L), dtype=np.int64)
        am[0, L - L // 8:] = 0
        spans = [(3, 3 + Nv, (L * 3) // 4), (L // 3, L // 3 + Nv, L - 2)]
        masks = []
        for b in range(B):
            a = torch.from_numpy(am[b])
            s, t, e = spans[b]
            masks.append(fn(attention_mask_2d=a, image_start_idx=s, text_start_idx=t, text_end_idx=e,
                            input_ids_shape=a.shape, dtype=a.dtype, device=a.device))
        m4 = ref.utils.stack_with_padding_2D_attention(masks)
        xt = torch.from_numpy(x)
        pos = torch.arange(L)[None]
        cos, sin = rot(xt, pos)
        y32, _ = attn(xt, (cos, sin), R.invert_mask_441(m4, torch.float32))
        a16 = Phi3Attention(cfg, layer_idx=0).eval()
        a16.load_state_dict(attn.state_dict())
        a16 = a16.to(torch.bfloat16)
        xb = xt.to(torch.bfloat16)
        cosb, sinb = rot(xb, pos)
        y16, _ = a16(xb, (cosb, sinb), R.invert_mask_441(m4, torch.bfloat16))
        save(f"attn_block_{tag}.npz", shapes=np.array(json.dumps(shapes)), am=am, spans=np.array(spans),
             mask_bits=gen.pack_mask_bits(m4.numpy()), mask_shape=np.array(m4.shape),
             cos=cos.numpy(), sin=sin.numpy(), y32=y32.numpy(), y16=y16.float().numpy())


def g_rope():
    """LongRoPE cos/sin as Phi-3.5 uses them (HF rope utils 'longrope'), short and long regimes."""
    from transformers import Phi3Config
    from transformers.models.phi3.modeling_phi3 import Phi3RotaryEmbedding
    rng = gen.rng_for("rope")
    short = (1.0 + rng.random(48) * 0.5).tolist()
    long = (1.0 + rng.random(48) * 30).tolist()
    cfg = Phi3Config(hidden_size=3072, num_attention_heads=32, max_position_embeddings=131072,
                     original_max_position_embeddings=4096,
                     rope_parameters={"rope_type": "longrope", "rope_theta": 10000.0, "short_factor": short,
                                      "long_factor": long, "original_max_position_embeddings": 4096})
    rot = Phi3RotaryEmbedding(cfg)
    x = torch.zeros(1, 1, 96)
    pos_s = torch.tensor([[0, 1, 2, 100, 654, 4095]])
    cs, ss = rot(x, pos_s)
    pos_l = torch.tensor([[0, 7, 4096, 5000]])
    cl, sl = rot(x, pos_l)
    save("rope_longrope.npz", short=np.array(short, dtype=np.float32), long=np.array(long, dtype=np.float32),
         pos_s=pos_s.numpy(), cos_s=cs.numpy(), sin_s=ss.numpy(), pos_l=pos_l.numpy(), cos_l=cl.numpy(), sin_l=sl.numpy(),
         attention_scaling=np.array(float(rot.attention_scaling)))


def g_patch_embed():
    """a3: SiglipVisionEmbeddings at AKI-4B dims (384 px, 1152 channels): row subset + checksums."""
    from transformers import SiglipVisionConfig
    import transformers.models.siglip.modeling_siglip as ms
    cfg = SiglipVisionConfig(hidden_size=1152, image_size=384, patch_size=14, num_hidden_layers=1,
                             num_attention_heads=16, intermediate_size=64)
    emb = ms.SiglipVisionEmbeddings(cfg).eval()
    shapes = load_filled(emb, 41)
    x = gen.rng_for("patch_embed").random((2, 3, 384, 384), dtype=np.float32) * 2 - 1
    y = emb(torch.from_numpy(x)).numpy()
    rows = np.array([0, 1, 26, 27, 364, 728])
    save("patch_embed_full.npz", shapes=np.array(json.dumps(shapes)), rows=rows, y_rows=y[:, rows],
         y_sum=np.array(float(y.astype(np.float64).sum())), y_abs=np.array(float(np.abs(y.astype(np.float64)).sum())))


def g_sft_collate():
    """8(f) #4: `batch_collate_pad` of the reference's SFT data pipeline on seeded ragged batches (inputs are regenerated from the
    seed by the tests; only the reference's outputs are stored)."""
    lu = R.load_sft_loader_utils()
    out = {}
    for i, (batch, padding, side, pad_id, max_length) in enumerate(gen.sft_cases()):
        res = lu.batch_collate_pad([{k: (torch.tensor(v) if (i % 2 and k == "input_ids") else v) for k, v in s.items()} for s in batch],
                                   padding, side, pad_id, max_length)
        for k, v in res.items():
            out[f"{k}_{i}"] = v.numpy()
    save("sft_collate.npz", n_cases=np.array(len(gen.sft_cases())), **out)


def g_train_losses():
    """a13 callers: the reference's LR schedule multipliers and the labels its two loss callables hand to the model (a stub model
    records them) - train/losses.py:10-40,83-151."""
    tl = R.load_train_losses()
    scheds, steps, pad, specials, ids, labels = gen.loss_cases()
    out = {}
    for i, sc in enumerate(scheds):
        opt = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=sc["lr"])
        sched = tl.get_cosine_schedule_with_warmup(opt, sc["lr"], sc["min_lr"], sc["num_warmup_steps"], sc["num_training_steps"], sc["num_cycles"])
        out[f"mult_{i}"] = np.array([sched.lr_lambdas[0](st) for st in steps], dtype=np.float64)
    seen = {}

    class Stub:
        special_token_ids = specials

        def __call__(self, **kw):
            seen.update(kw)
            return (torch.tensor(1.5),)

    import contextlib
    tok = type("Tok", (), {"pad_token_id": pad})()
    tl.NextTokenPrediction()(Stub(), tok, None, torch.from_numpy(ids.copy()), torch.ones_like(torch.from_numpy(ids)), contextlib.nullcontext)
    out["ntp_labels"] = seen["labels"].numpy().copy()
    seen.clear()
    tl.SupervisedPrediction()(Stub(), tok, None, torch.from_numpy(ids.copy()), torch.from_numpy(labels.copy()),
                              torch.ones_like(torch.from_numpy(ids)), contextlib.nullcontext)
    out["sft_labels"] = seen["labels"].numpy().copy()
    save("train_losses.npz", **out)


def g_tiny_generate(ref, n_new=8):
    """f1: greedy continuation produced by the REFERENCE (src/aki.py:136-209 + src/aki_generation.py:36-86).  The patched
    `_update_model_kwargs_for_generation` needs private API of transformers 4.41.2, so the reference's `generate` cannot be driven
    under the installed 5.x; its semantics can: after the MMA prefill every new token attends to all earlier positions (the
    all-ones mask of aki_generation.py:58-62) at position = its index, and the prompt's rows keep the MMA mask they were cached
    under.  A full `AKI.forward` of the reference over prompt + tokens-so-far computes exactly that last row (the mask of the
    prompt part is unchanged as long as no generated token is <image> / <|assistant|>, asserted below), so the continuation is
    produced by repeated full forwards of the imported reference, one sample at a time - the reference's generate is batch-1
    in effect.  Stored: token ids, the top-1 / top-2 margin per step (near-ties are data, not failures), and the
    chosen-row logits on a column subset."""
    model, _ = build_tiny(ref)
    lang_x, am, _, vision_x = tiny_batch()
    T = gen.TINY
    cols = np.unique(np.concatenate((np.arange(0, 32013, 997), [1, 2, 32000, 32001, 32007, 32010, 32011, 32012])))
    B = lang_x.shape[0]
    toks = np.zeros((B, n_new), dtype=np.int64)
    margin = np.zeros((B, n_new), dtype=np.float32)
    rows = np.zeros((B, n_new, len(cols)), dtype=np.float32)
    for b in range(B):
        ids = torch.from_numpy(lang_x[b, : int(am[b].sum())])[None]
        vx = torch.from_numpy(vision_x[b:b + 1])
        for t in range(n_new):
            logits = model(vx, ids, attention_mask=torch.ones_like(ids)).logits[0, -1]
            top2 = logits.topk(2)
            nxt = int(top2.indices[0])
            assert nxt not in (T["media_token_id"], 32001), "a generated special token would change the reference's mask"
            toks[b, t], margin[b, t] = nxt, float(top2.values[0] - top2.values[1])
            rows[b, t] = logits[cols].numpy()
            ids = torch.cat([ids, torch.tensor([[nxt]])], dim=1)
    print("generate margins (top1 - top2):", np.round(margin, 5).tolist())
    save("tiny_generate.npz", tokens=toks, margin=margin, logit_cols=cols, logits=rows)


def main():
    ref = R.load_reference()
    g_train_losses()
    g_sft_collate()
    g_mask(ref)
    g_decoupled(ref)
    g_rope()
    g_attn_block(ref)
    g_perceiver(ref)
    g_patch_embed()
    g_tiny_e2e(ref)
    g_tiny_grads(ref)
    g_tiny_generate(ref)


if __name__ == "__main__":
    main()
