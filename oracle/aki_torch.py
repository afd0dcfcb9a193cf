"""TEST INFRASTRUCTURE ONLY - torch (CPU, fp32, eager) restatement of the reference's AKI forward.

Same algorithm and the same function / parameter names as oracle/aki_oracle.py (numpy), written with torch ops so that
  * it is differentiable: torch autograd over this file is the checker for the training-step backward kernels
    (the reference's backward IS autograd over its eager forward: train/train_utils.py:242-266), and
  * it runs multi-threaded on the host cores: bench.py's `cpu_baseline` leg times it ("the build's eager
    restatement run on the GPU box's host cores").
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import it; the product (aki_amd/) never does.

Pinning: tests/test_oracle_golden.py checks this file against the golden vectors captured from the imported reference
(tests/golden/make_golden.py): tiny end-to-end logits / loss, the attention block, the Perceiver, and - for backward -
parameter gradients of the reference's own loss.backward() on the tiny model (tiny_grads.npz).
Integer bookkeeping (mask construction, splice indices) is shared with aki_oracle.py, which is pinned bit-exactly.

Reference lines followed by each function are cited in its docstring (same citations as aki_oracle.py).
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F

import aki_oracle as O

ASSISTANT_TOKEN_ID = O.ASSISTANT_TOKEN_ID


def _sub(p: Dict[str, torch.Tensor], prefix: str) -> Dict[str, torch.Tensor]:
    return {k[len(prefix):]: v for k, v in p.items() if k.startswith(prefix)}


def layer_norm(x, w, b, eps=1e-5):
    return F.layer_norm(x, (x.shape[-1],), w, b, eps)


def rms_norm(x, w, eps=1e-5):
    """Phi3RMSNorm, HF:phi3/modeling_phi3.py:266-284."""
    v = x.float().pow(2).mean(-1, keepdim=True)
    return w * (x.float() * torch.rsqrt(v + eps)).to(x.dtype)


def linear(x, w, b=None):
    return F.linear(x, w, b)


# ---- BASELINE configs[4]: e4m3 fake quantisation at the build's quantisation points -----------------------------------
# No reference behaviour exists for fp8 (the reference has no fp8 path): this block is a BUILD-DEFINED oracle.  It restates what
# the product computes - aki_amd/csrc/fp8_quant.hip (one dynamic scale per row: per token for activations, per output feature
# for nn.Linear weights, s = max(amax, 1e-12) / 448, q = RNE_e4m3(clamp(y / s))) and the fp8 GEMM (f32 accumulation of the
# e4m3 products, accumulator times s_x[token] * s_w[feature]) - with torch's own float8_e4m3fn cast, so that the HIP fp8 path
# has something to be compared with at bf16-level bars instead of "within the format's noise of the fp32 model".
FP8_MAX = 448.0


def quant_rows_e4m3(y):
    """-> (q: e4m3 VALUES as f32 [rows.., cols], s: f32 [rows.., 1]) with y ~= q * s."""
    yf = y.float()
    s = yf.abs().amax(-1, keepdim=True).clamp(min=1e-12) * (1.0 / FP8_MAX)
    q = (yf * (1.0 / s)).clamp(-FP8_MAX, FP8_MAX).to(torch.float8_e4m3fn).float()
    return q, s


def linear_e4m3(x, w, b=None):
    """F.linear with both operands fake-quantised per row to e4m3; products and sums in f32, one rounding to x.dtype at the end
    (the GEMM epilogue's)."""
    wq, ws = quant_rows_e4m3(w.detach())
    xq, xs = quant_rows_e4m3(x)
    y = F.linear(xq, wq) * xs * ws.reshape(-1)
    if b is not None:
        y = y + b.float()
    return y.to(x.dtype)


def _lin(fp8, which):
    """The linear used for projection `which` under the fp8 configuration dict (None = everything in the model dtype).
    Keys mirror Phi3ForCausalLM.enable_fp8(head=..., residual_writers=...)."""
    if not fp8:
        return linear
    if which in ("o", "down") and not fp8.get("residual_writers", True):
        return linear
    if which == "head" and not fp8.get("head", True):
        return linear
    return linear_e4m3


# ---- a5 / a12 -------------------------------------------------------------------------------------
def decoupled_embedding(ids, weight, additional_weight, max_original_id):
    """src/helpers.py:445-484."""
    if additional_weight is None:
        return F.embedding(ids, weight)
    hi = ids > max_original_id
    low = F.embedding(torch.where(hi, torch.zeros_like(ids), ids), weight)
    add = F.embedding(torch.where(hi, ids - max_original_id - 1, torch.zeros_like(ids)), additional_weight)
    return torch.where(hi[..., None], add, low)


def decoupled_linear(x, weight, bias, add_weight, add_bias, max_original_id, lin=linear):
    """src/helpers.py:594-603."""
    out = lin(x, weight, bias)[..., : max_original_id + 1]
    if add_weight is not None:
        out = torch.cat((out, lin(x, add_weight, add_bias)), -1)
    return out


# ---- a6-a8: splice ---------------------------------------------------------------------------------
def prepare_inputs_multi_image(vision_tokens, lang_x, attention_mask, lang_embeds, media_token_id, pad_token_id, num_tokens_per_vis):
    """BUILD-DEFINED (parity unpinned: the reference raises on a second image, src/vlm.py:547-554 - BASELINE configs[3] needs four).  The
    rule the product implements (csrc/aux_kernels.hip splice_table_kernel), restated: the k-th `<image>` placeholder at prompt index t_k is
    replaced by that image's Nv vision tokens, so it starts at E_k = t_k + k (Nv - 1) of the spliced stream; rows [E_k, E_k + Nv) additionally
    see columns [E_k + Nv, text_end) with text_end = spliced index of the first <|assistant|> token + 1 (src/vlm.py:556-564 applied per image);
    everything else is causal, padded columns are masked (src/vlm.py:434-438).  Right padding, no labels."""
    lx, am = lang_x.cpu().numpy(), attention_mask.cpu().numpy()
    B, Nv = lx.shape[0], num_tokens_per_vis
    embeds, masks, spans, m1d = [], [], [], []
    for i in range(B):
        img = [int(t) for t in np.where(lx[i] == media_token_id)[0]]
        q = np.where(lx[i] == O.ASSISTANT_TOKEN_ID)[0]
        q = int(q[0]) if len(q) else 0
        parts, mparts, prev = [], [], 0
        for k, t in enumerate(img):
            parts += [lang_embeds[i][prev:t], vision_tokens[i][k]]
            mparts += [am[i][prev:t], np.ones(Nv, dtype=am.dtype)]
            prev = t + 1
        parts.append(lang_embeds[i][prev:])
        mparts.append(am[i][prev:])
        e, a = torch.cat(parts, 0), np.concatenate(mparts, 0)
        n = a.shape[0]
        text_end = q + sum(1 for t in img if t < q) * (Nv - 1) + 1
        rects = []
        for k, t in enumerate(img):
            st = t + k * (Nv - 1)
            r = (min(st, n), min(st + Nv, n), min(st + Nv, n), min(text_end, n))
            rects.append(r if r[1] > r[0] and r[3] > r[2] else (0, 0, 0, 0))
        embeds.append(e)
        masks.append(O.mask_from_spans(a, rects))
        spans.append(rects)
        m1d.append(a)
    Lmax = max(e.shape[0] for e in embeds)
    rows = [e if e.shape[0] == Lmax else torch.cat((e, torch.full((Lmax - e.shape[0], e.shape[1]), float(pad_token_id), dtype=e.dtype)), 0) for e in embeds]
    return {"inputs_embeds": torch.stack(rows), "attention_mask": torch.from_numpy(O.stack_with_padding_2d_attention(masks)), "labels": None,
            "spans": spans, "mask_1d": O.stack_with_padding(m1d, padding_value=0, padding_side="right"), "lengths": [e.shape[0] for e in embeds]}


def prepare_inputs_for_forward(vision_tokens, lang_x, attention_mask, labels, lang_embeds, media_token_id, pad_token_id,
                               num_tokens_per_vis, padding_side="right"):
    """src/vlm.py:445-603 (no KV cache).  Index logic and the dense mask come from the numpy oracle (bit-exact, pinned);
    the embedding splice is done here with torch ops so gradients reach vision tokens and text embeddings."""
    lx = lang_x.cpu().numpy()
    am = attention_mask.cpu().numpy()
    lb = None if labels is None else labels.cpu().numpy()
    B = lx.shape[0]
    dummy = np.zeros(lx.shape + (1,), dtype=np.float32)
    dvt = None if vision_tokens is None else np.zeros(tuple(vision_tokens.shape[:3]) + (1,), dtype=np.float32)
    prep = O.prepare_inputs_for_forward(dvt, lx, am, lb, dummy, media_token_id, pad_token_id, num_tokens_per_vis, padding_side)
    embeds = []
    for i in range(B):
        img = np.where(lx[i] == media_token_id)[0]
        if len(img) == 0:
            embeds.append(lang_embeds[i])
        else:
            k = int(img[0])
            embeds.append(torch.cat((lang_embeds[i][:k], vision_tokens[i][0], lang_embeds[i][k + 1:]), 0))
    Lmax = max(e.shape[0] for e in embeds)
    rows = []
    for e in embeds:                                   # src/utils.py:62-96: pad with the scalar pad_token_id
        padn = Lmax - e.shape[0]
        if padn:
            fill = torch.full((padn, e.shape[1]), float(pad_token_id), dtype=e.dtype)
            e = torch.cat((e, fill), 0) if padding_side == "right" else torch.cat((fill, e), 0)
        rows.append(e)
    return {"inputs_embeds": torch.stack(rows), "attention_mask": torch.from_numpy(prep["attention_mask"]),
            "labels": None if labels is None else torch.from_numpy(prep["labels"]), "spans": prep["spans"],
            "mask_1d": prep["mask_1d"], "lengths": prep["lengths"]}


# ---- a4: Perceiver ---------------------------------------------------------------------------------
def feed_forward(x, ln_w, ln_b, w1, w2):
    """src/helpers.py:32-39."""
    return F.linear(F.gelu(F.linear(layer_norm(x, ln_w, ln_b), w1)), w2)


def perceiver_attention(x, latents, p, heads=8, dim_head=64):
    """src/helpers.py:76-102."""
    x = layer_norm(x, p["norm_media.weight"], p["norm_media.bias"])
    latents = layer_norm(latents, p["norm_latents.weight"], p["norm_latents.bias"])
    q = F.linear(latents, p["to_q.weight"])
    kv = F.linear(torch.cat((x, latents), -2), p["to_kv.weight"])
    inner = heads * dim_head
    k, v = kv[..., :inner], kv[..., inner:]

    def split(t):
        b, T, n, _ = t.shape
        return t.reshape(b, T, n, heads, dim_head).permute(0, 3, 1, 2, 4)

    q, k, v = split(q) * dim_head ** -0.5, split(k), split(v)
    sim = q @ k.transpose(-1, -2)
    sim = sim - sim.amax(-1, keepdim=True).detach()
    out = sim.softmax(-1) @ v
    b, h, T, n, d = out.shape
    return F.linear(out.permute(0, 2, 3, 1, 4).reshape(b, T, n, h * d), p["to_out.weight"])


def perceiver_resampler(x, p, depth=6, heads=8, dim_head=64):
    """src/helpers.py:170-199."""
    b, T, Fr, v, D = x.shape
    x = x.reshape(b, T, Fr * v, D)
    lat = p["latents"].expand(b, T, *p["latents"].shape)
    for l in range(depth):
        lat = perceiver_attention(x, lat, _sub(p, f"layers.{l}.0."), heads, dim_head) + lat
        lat = feed_forward(lat, p[f"layers.{l}.1.0.weight"], p[f"layers.{l}.1.0.bias"], p[f"layers.{l}.1.1.weight"],
                           p[f"layers.{l}.1.3.weight"]) + lat
    out = layer_norm(lat, p["norm.weight"], p["norm.bias"])
    if "projection.weight" in p:
        out = F.linear(out, p["projection.weight"], p["projection.bias"])
    return out


# ---- a9-a11: Phi-3 with the 4.41.2 mask hand-off -----------------------------------------------------
def invert_mask_441(mask01, dtype=torch.float32):
    """transformers==4.41.2 _prepare_4d_causal_attention_mask on a 4-D 0/1 mask."""
    inv = 1.0 - mask01.to(dtype)
    return inv.masked_fill(inv.bool(), torch.finfo(dtype).min)


def rope_cos_sin(position_ids, head_dim, theta=10000.0):
    cos, sin = O.rope_cos_sin(np.asarray(position_ids), head_dim, theta)
    return torch.from_numpy(cos), torch.from_numpy(sin)


def rotate_half(x):
    h = x.shape[-1] // 2
    return torch.cat((-x[..., h:], x[..., :h]), -1)


def phi3_attention(x, w_qkv, w_o, cos, sin, add_mask, n_heads, fp8=None):
    """HF:phi3/modeling_phi3.py:145-167,170-197,218-263.  fp8: see _lin (build-defined; the attention core itself stays in x.dtype)."""
    B, L, d = x.shape
    Dh = d // n_heads
    qkv = _lin(fp8, "qkv")(x, w_qkv)
    q, k, v = (t.reshape(B, L, n_heads, Dh).transpose(1, 2) for t in (qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:]))
    cu, su = cos[:, None].to(x.dtype), sin[:, None].to(x.dtype)
    q = q * cu + rotate_half(q) * su
    k = k * cu + rotate_half(k) * su
    s = (q @ k.transpose(-1, -2)) * Dh ** -0.5 + add_mask
    p = s.float().softmax(-1).to(x.dtype)
    o = (p @ v).transpose(1, 2).reshape(B, L, d)
    return _lin(fp8, "o")(o, w_o)


def phi3_mlp(x, w_gate_up, w_down, fp8=None):
    if fp8 and x.dtype != torch.float32:
        # the product's gate_up epilogue applies SwiGLU to the f32 accumulators and rounds once; staying in f32 up to that
        # rounding keeps the low-precision yardstick run from adding a rounding the kernel does not have
        up = linear_e4m3(x.float(), w_gate_up)
        gate, u = up.chunk(2, -1)
        a = (u * F.silu(gate)).to(x.dtype)
    else:
        up = _lin(fp8, "gate_up")(x, w_gate_up)
        gate, u = up.chunk(2, -1)
        a = u * F.silu(gate)
    return _lin(fp8, "down")(a, w_down)


def phi3_decoder_layer(h, p, cos, sin, add_mask, n_heads, eps=1e-5, fp8=None):
    x = rms_norm(h, p["input_layernorm.weight"], eps)
    h = h + phi3_attention(x, p["self_attn.qkv_proj.weight"], p["self_attn.o_proj.weight"], cos, sin, add_mask, n_heads, fp8)
    x = rms_norm(h, p["post_attention_layernorm.weight"], eps)
    return h + phi3_mlp(x, p["mlp.gate_up_proj.weight"], p["mlp.down_proj.weight"], fp8)


def phi3_lm_forward(inputs_embeds, mask01_4d, p, n_layers, n_heads, max_original_id, theta=10000.0, eps=1e-5, fp8=None, rope=None):
    """fp8 (None or {"head": bool, "residual_writers": bool}): BASELINE configs[4], build-defined - the decoder's projections
    (and optionally the head) on e4m3 fake-quantised operands at exactly the product's quantisation points: after each fused
    RMSNorm (qkv, gate_up, head), on the attention output (o_proj) and on the SwiGLU output (down_proj)."""
    B, L, d = inputs_embeds.shape
    if rope is None:
        cos, sin = rope_cos_sin(np.arange(L)[None], d // n_heads, theta)
    else:   # LongRoPE (HF modeling_rope_utils: inv_freq = 1 / (ext_factors * theta^(2i/d)), cos / sin scaled): {"ext_factors", "attention_scaling"}
        c_, s_ = O.rope_cos_sin(np.arange(L)[None], d // n_heads, theta, np.asarray(rope["ext_factors"], dtype=np.float32),
                                float(rope.get("attention_scaling", 1.0)))
        cos, sin = torch.from_numpy(c_), torch.from_numpy(s_)
    add = invert_mask_441(mask01_4d, inputs_embeds.dtype)
    h = inputs_embeds
    for l in range(n_layers):
        h = phi3_decoder_layer(h, _sub(p, f"model.layers.{l}."), cos, sin, add, n_heads, eps, fp8)
    h = rms_norm(h, p["model.norm.weight"], eps)
    return decoupled_linear(h, p["lm_head.weight"], p.get("lm_head.bias"), p.get("lm_head.additional_fc.weight"),
                            p.get("lm_head.additional_fc.bias"), max_original_id, _lin(fp8, "head"))


def causal_lm_loss(logits, labels, ignore_index=-100):
    """HF shifted CE (mean over non-ignored targets, fp32), as used by train/losses.py:83-116."""
    lg = logits[:, :-1].float().reshape(-1, logits.shape[-1])
    return F.cross_entropy(lg, labels[:, 1:].reshape(-1), ignore_index=ignore_index)


# ---- a2/a3: SigLIP ---------------------------------------------------------------------------------
def siglip_patch_embed(pixels, w, b, pos_emb):
    """HF:siglip/modeling_siglip.py:175-185."""
    return F.conv2d(pixels, w, b, stride=w.shape[-1]).flatten(2).transpose(1, 2) + pos_emb[None]


def siglip_encoder_layer(h, p, n_heads, eps=1e-6):
    """HF:siglip/modeling_siglip.py:250-357."""
    N, L, E = h.shape
    Dh = E // n_heads
    x = layer_norm(h, p["layer_norm1.weight"], p["layer_norm1.bias"], eps)
    q, k, v = (F.linear(x, p[f"self_attn.{n}_proj.weight"], p[f"self_attn.{n}_proj.bias"]).reshape(N, L, n_heads, Dh).transpose(1, 2)
               for n in ("q", "k", "v"))
    a = ((q @ k.transpose(-1, -2)) * Dh ** -0.5).softmax(-1) @ v
    h = h + F.linear(a.transpose(1, 2).reshape(N, L, E), p["self_attn.out_proj.weight"], p["self_attn.out_proj.bias"])
    x = layer_norm(h, p["layer_norm2.weight"], p["layer_norm2.bias"], eps)
    x = F.linear(F.gelu(F.linear(x, p["mlp.fc1.weight"], p["mlp.fc1.bias"]), approximate="tanh"), p["mlp.fc2.weight"], p["mlp.fc2.bias"])
    return h + x


def siglip_interpolate_pos(pos_emb, grid):
    """HF:siglip/modeling_siglip.py `interpolate_pos_encoding`: the learned (g0*g0, E) table resampled bicubically
    (align_corners=False) to grid x grid.  The reference never takes this path (it only runs the tower's native 384 px,
    src/vlm.py:202-203); BASELINE's 336 px metric resolution does - pinned against transformers' own method in
    tests/test_model_gpu.py::test_full_width_siglip_tower_vs_transformers."""
    g0 = int(round(pos_emb.shape[0] ** 0.5))
    if grid == g0:
        return pos_emb
    t = pos_emb.float().reshape(1, g0, g0, -1).permute(0, 3, 1, 2)
    t = F.interpolate(t, size=(grid, grid), mode="bicubic", align_corners=False)
    return t.permute(0, 2, 3, 1).reshape(grid * grid, -1).to(pos_emb.dtype)


def siglip_vision_forward(pixels, p, n_layers, n_heads, eps=1e-6):
    w = p["embeddings.patch_embedding.weight"]
    pos = siglip_interpolate_pos(p["embeddings.position_embedding.weight"], pixels.shape[-1] // w.shape[-1])
    h = siglip_patch_embed(pixels, w, p["embeddings.patch_embedding.bias"], pos)
    for l in range(n_layers):
        h = siglip_encoder_layer(h, _sub(p, f"encoder.layers.{l}."), n_heads, eps)
    return layer_norm(h, p["post_layernorm.weight"], p["post_layernorm.bias"], eps)


# ---- a1 ----------------------------------------------------------------------------------------------
def aki_forward(p: Dict[str, torch.Tensor], cfg: Dict, vision_x, lang_x, attention_mask, labels=None):
    """AKI.forward (src/aki.py:65-134) in fp32 eager torch; the vision tower runs under no_grad (src/vlm.py:184-207)."""
    b, T, Fr = vision_x.shape[:3]
    with torch.no_grad():
        feats = siglip_vision_forward(vision_x.reshape((b * T * Fr,) + tuple(vision_x.shape[3:])), _sub(p, "vision_encoder."),
                                      cfg["vis_layers"], cfg["vis_heads"])
    feats = feats.reshape(b, T, Fr, feats.shape[1], feats.shape[2])
    vt = perceiver_resampler(feats, _sub(p, "vision_tokenizer."), cfg.get("perc_depth", 6), cfg.get("perc_heads", 8),
                             cfg.get("perc_dim_head", 64))
    lm = _sub(p, "lang_model.")
    emb = decoupled_embedding(lang_x, lm["model.embed_tokens.weight"], lm.get("model.embed_tokens.additional_embedding.weight"),
                              cfg["max_original_id"])
    if cfg.get("multi_image"):      # build-defined (BASELINE configs[3]); the reference's behaviour is the branch below, which raises
        assert labels is None
        prep = prepare_inputs_multi_image(vt, lang_x, attention_mask, emb, cfg["media_token_id"], cfg["pad_token_id"], cfg["num_vision_tokens"])
    else:
        prep = prepare_inputs_for_forward(vt, lang_x, attention_mask, labels, emb, cfg["media_token_id"], cfg["pad_token_id"],
                                          cfg["num_vision_tokens"], "right")
    logits = phi3_lm_forward(prep["inputs_embeds"], prep["attention_mask"], lm, cfg["lm_layers"], cfg["lm_heads"],
                             cfg["max_original_id"], cfg.get("rope_theta", 10000.0), cfg.get("rms_eps", 1e-5), cfg.get("fp8"), cfg.get("rope"))
    loss = causal_lm_loss(logits, prep["labels"]) if labels is not None else None
    return {"logits": logits, "loss": loss, "prep": prep, "vision_tokens": vt}
