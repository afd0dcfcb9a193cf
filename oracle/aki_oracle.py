"""CPU oracle for the AKI modality-mutual-attention (MMA) forward path.  TEST INFRASTRUCTURE ONLY.

This file is a plain-numpy restatement of the reference's algorithm for the hot path of SURVEY.md
section 8.  It exists to CHECK the HIP path; it is never the thing shipped or measured.  Only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.
Nothing in ``aki_amd/`` imports it and the product path fails loudly when the HIP library is absent.

Parity pinning: the reference has no tests or golden vectors of its own.  The
oracle is pinned against outputs of the reference itself, imported in the build container by
``tests/golden/make_golden.py`` (first-party ``src/*.py``) together with the installed
``transformers`` Phi-3 / SigLIP building blocks that the reference delegates its attention
arithmetic to (third-party: ``transformers==4.41.2`` pinned in ``codes/setup.py:9`` plus the hub
``modeling_phi3.py`` of ``microsoft/Phi-3.5-mini-instruct`` via ``trust_remote_code``; neither is
vendored under /root/reference).  ``tests/test_oracle_golden.py`` checks every function below
against those committed vectors.  Build-defined paths with no reference behaviour (multi-image
masks, 336 px position-embedding interpolation, fp8) are marked "parity unpinned" where they appear.

Citations: ``src/...`` = /root/reference/codes/open_flamingo/src/..., ``HF:`` = installed
transformers 5.15.0 ``models/...`` (same arithmetic as the pinned 4.41.2 for these functions,
except the 4-D mask hand-off restated in :func:`invert_mask_441`).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

F32_MIN = float(np.finfo(np.float32).min)
BF16_MIN = -3.3895313892515355e38  # torch.finfo(torch.bfloat16).min
ASSISTANT_TOKEN_ID = 32001  # hard-coded in src/vlm.py:490-496


# ----------------------------------------------------------------------------------------------
# numeric helpers
# ----------------------------------------------------------------------------------------------
def bf16_round(x: np.ndarray) -> np.ndarray:
    """fp32 -> bf16 (round-to-nearest-even) -> fp32.  NaN stays NaN."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    u = x.view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    out = r.astype(np.uint32).view(np.float32).reshape(x.shape)
    nan = np.isnan(x)
    if nan.any():
        out = np.where(nan, np.float32(np.nan), out)
    return out


def to_bf16_bits(x: np.ndarray) -> np.ndarray:
    """fp32 -> uint16 bf16 bit patterns (RNE)."""
    return (bf16_round(x).view(np.uint32) >> 16).astype(np.uint16)


def from_bf16_bits(b: np.ndarray) -> np.ndarray:
    return (b.astype(np.uint32) << 16).view(np.float32)


def softmax(x: np.ndarray, axis: int = -1) -> np.ndarray:
    m = np.max(x, axis=axis, keepdims=True)
    e = np.exp(x - m)
    return e / np.sum(e, axis=axis, keepdims=True)


def _erf(x):
    try:
        from scipy.special import erf
        return erf(x)
    except Exception:  # pragma: no cover
        return np.vectorize(math.erf)(x)


def gelu_erf(x):
    """torch.nn.GELU() default (exact erf form) - src/helpers.py:37."""
    return (0.5 * x * (1.0 + _erf(x / np.sqrt(2.0)))).astype(x.dtype)


def gelu_tanh(x):
    """gelu_pytorch_tanh - HF:siglip/configuration_siglip.py hidden_act."""
    c = np.sqrt(2.0 / np.pi)
    return (0.5 * x * (1.0 + np.tanh(c * (x + 0.044715 * x ** 3)))).astype(x.dtype)


def silu(x):
    return (x / (1.0 + np.exp(-x))).astype(x.dtype)


def layer_norm(x, w, b, eps=1e-5):
    mu = x.mean(-1, keepdims=True)
    var = ((x - mu) ** 2).mean(-1, keepdims=True)
    return ((x - mu) / np.sqrt(var + eps) * w + b).astype(x.dtype)


def rms_norm(x, w, eps=1e-5):
    """Phi3RMSNorm - HF:phi3/modeling_phi3.py:266-284 (fp32 statistics)."""
    xf = x.astype(np.float32)
    var = (xf * xf).mean(-1, keepdims=True)
    return (w * (xf / np.sqrt(var + eps))).astype(x.dtype)


def linear(x, w, b=None):
    y = x @ w.T
    if b is not None:
        y = y + b
    return y


# ----------------------------------------------------------------------------------------------
# a7: modality-mutual mask (integer, bit-exact)
# ----------------------------------------------------------------------------------------------
def make_modality_mutual_mask(attention_mask_1d: np.ndarray, image_start_idx: int, text_start_idx: int,
                              text_end_idx: int) -> np.ndarray:
    """``VLMWithLanguageStream._make_modality_mutual_mask`` (src/vlm.py:410-443), restated step by step.

    Returns int64 ``(1, n, n)`` 0/1.  Python slice semantics for the unlock rectangle are kept
    (negative or reversed indices behave as numpy/torch slicing does).
    """
    am = np.asarray(attention_mask_1d)
    n = am.shape[0]
    col = np.arange(n)
    mask = (col[None, :] < (col + 1)[:, None]).astype(np.int64)          # :424-426 tril
    mask[image_start_idx:text_start_idx, text_start_idx:text_end_idx] = 1  # :429
    inv = 1.0 - am.astype(np.float32)                                      # :434-436
    masked_cols = inv.astype(bool)
    mask = np.where(masked_cols[None, :], 0, mask)                         # :438
    return mask[None, :, :].astype(np.int64)


def mma_visible(r, c, valid_cols, rects: Sequence[Tuple[int, int, int, int]]):
    """Closed form used by the HIP kernels:
    ``visible(r,c) = valid(c) and (c <= r or exists k: r0_k <= r < r1_k and c0_k <= c < c1_k)``
    with ``rects = [(row_lo,row_hi,col_lo,col_hi), ...]`` (see :func:`clamp_span`).  One rectangle
    per sample is the reference's behaviour; several are build-defined (parity unpinned)."""
    vis = c <= r
    for (r0, r1, c0, c1) in rects:
        vis = vis | ((r >= r0) & (r < r1) & (c >= c0) & (c < c1))
    return vis & valid_cols[c]


def mask_from_spans(attention_mask_1d: np.ndarray, rects: Sequence[Tuple[int, int, int, int]]) -> np.ndarray:
    am = np.asarray(attention_mask_1d).astype(bool)
    n = am.shape[0]
    r = np.arange(n)[:, None]
    c = np.arange(n)[None, :]
    return mma_visible(r, c, am, rects).astype(np.int64)[None]


def clamp_span(n: int, image_start: int, text_start: int, text_end: int) -> Tuple[int, int, int, int]:
    """The unlock rectangle of ``mask[image_start:text_start, text_start:text_end] = 1`` (src/vlm.py:429)
    under Python slice normalisation, as (row_lo,row_hi,col_lo,col_hi); (0,0,0,0) when empty."""
    rs, re_, _ = slice(image_start, text_start).indices(n)
    cs, ce, _ = slice(text_start, text_end).indices(n)
    if re_ <= rs or ce <= cs:
        return (0, 0, 0, 0)
    return (rs, re_, cs, ce)


def mask_to_table(mask01: np.ndarray, max_rects: int = 8):
    """Inverse of :func:`mask_from_spans` for the reference's LM hand-off type: one sample's dense (L, L) 0/1 mask
    (what `_prepare_inputs_for_forward` returns, src/vlm.py:589-603, after stacking, src/utils.py:99-108) ->
    (rects [(row_lo,row_hi,col_lo,col_hi)...], valid_cols bool [L], seq_len).  Build-defined helper (the reference has no
    such function): it is pinned by the round trip through the reference's own masks - `mask_from_spans(valid, rects)`
    restricted to rows < seq_len must give back `mask01` for every golden mask case (tests/test_oracle_golden.py).
    Rules: valid(c) = any row sees c; seq_len = last non-empty row + 1; consecutive rows with the same right-of-diagonal
    interval [lo, hi) form one rectangle, a row whose interval starts at r+1 (clipped by the diagonal) joins the run
    above it.  Returns None when more than max_rects rectangles would be needed."""
    m = np.asarray(mask01) != 0
    L = m.shape[0]
    valid = m.any(axis=0)
    rows_any = m.any(axis=1)
    seq_len = int(np.nonzero(rows_any)[0].max()) + 1 if rows_any.any() else 0
    rects, run = [], None      # run = [start, lo, hi]
    for r in range(L):
        right = np.nonzero(m[r, r + 1:])[0]
        has = len(right) > 0
        lo, hi = (int(right[0]) + r + 1, int(right[-1]) + r + 2) if has else (0, 0)
        joins = run is not None and has and hi == run[2] and (lo == run[1] or (lo == r + 1 and run[1] <= r))
        if run is not None and not joins:
            rects.append((run[0], r, run[1], run[2]))
            run = None
        if has and run is None:
            run = [r, lo, hi]
    if run is not None:
        rects.append((run[0], L, run[1], run[2]))
    if len(rects) > max_rects:
        return None
    return rects, valid, seq_len


# ----------------------------------------------------------------------------------------------
# a8: padding / stacking
# ----------------------------------------------------------------------------------------------
def stack_with_padding(tensors: List[np.ndarray], padding_value=0, padding_side="right") -> np.ndarray:
    """src/utils.py:62-96."""
    mx = max(t.shape[0] for t in tensors)
    out = []
    for t in tensors:
        padshape = (mx - t.shape[0],) + tuple(t.shape[1:])
        pad = np.full(padshape, padding_value, dtype=t.dtype)
        out.append(np.concatenate((t, pad), 0) if padding_side == "right" else np.concatenate((pad, t), 0))
    return np.stack(out)


def stack_with_padding_2d_attention(tensors: List[np.ndarray]) -> np.ndarray:
    """src/utils.py:99-108: zero-pad bottom/right of each (1,n,n) mask to the max size, then stack."""
    mx = max(t.shape[1] for t in tensors)
    out = []
    for t in tensors:
        a = t.shape[-1]
        out.append(np.pad(t, ((0, 0), (0, mx - a), (0, mx - a))))
    return np.stack(out)


# ----------------------------------------------------------------------------------------------
# 8(f) #4: SFT collate (train/sft_data_utils/loader_utils.py)
# ----------------------------------------------------------------------------------------------
SFT_IGNORE_INDEX = -100   # train/sft_data_utils/templates/templates.py IGNORE_INDEX


def sft_pad_trunc(x, padding: str, padding_side: str, pad_value: int, max_length) -> np.ndarray:
    """`_pad_trunc`, loader_utils.py:11-50: every sample to one length - "longest": the longest sample's; "max_length": the
    given one; longer samples keep their FIRST max_length tokens, shorter ones are padded left or right."""
    lengths = [len(s) for s in x]
    if padding == "longest":
        max_length = max(lengths)                                   # :30-31 (the caller's limit is dropped here, as in the reference)
    rows = []
    for s, n in zip(x, lengths):
        s = [int(v) for v in (s.tolist() if hasattr(s, "tolist") else s)]
        if n >= max_length:
            rows.append(s[:max_length])                             # :38-40
        else:
            pads = [pad_value] * (max_length - n)
            rows.append(s + pads if padding_side == "right" else pads + s)   # :42-47
    return np.asarray(rows, dtype=np.int64)


def sft_batch_collate_pad(batch, padding: str, padding_side: str, pad_token_id: int, max_length):
    """`batch_collate_pad`, loader_utils.py:53-91 (note `max_length + 1` for the BOS token, :78-82)."""
    if padding != "max_length":
        max_length = max_length or int(1e12)
    cols = {k: [s[k] for s in batch] for k in ("input_ids", "labels", "attention_mask")}
    pads = {"input_ids": pad_token_id, "labels": SFT_IGNORE_INDEX, "attention_mask": 0}
    return {k: sft_pad_trunc(v, padding, padding_side, pads[k], max_length + 1) for k, v in cols.items()}


# ----------------------------------------------------------------------------------------------
# a5 / a12: decoupled embedding + lm_head
# ----------------------------------------------------------------------------------------------
def decoupled_embedding(ids: np.ndarray, weight: np.ndarray, additional_weight: Optional[np.ndarray],
                        max_original_id: int) -> np.ndarray:
    """src/helpers.py:445-484."""
    ids = np.asarray(ids)
    if additional_weight is None:
        return weight[ids]
    hi = ids > max_original_id
    low_ids = np.where(hi, 0, ids)
    full = weight[low_ids].copy()
    if hi.any():
        full[hi] = additional_weight[ids[hi] - max_original_id - 1]
    return full


def decoupled_linear(x, weight, bias, add_weight, add_bias, max_original_id: int):
    """src/helpers.py:594-603."""
    out = linear(x, weight, bias)[..., : max_original_id + 1]
    if add_weight is not None:
        out = np.concatenate((out, linear(x, add_weight, add_bias)), -1)
    return out


# ----------------------------------------------------------------------------------------------
# a6: splice vision tokens into the language stream
# ----------------------------------------------------------------------------------------------
def prepare_inputs_for_forward(vision_tokens: Optional[np.ndarray], lang_x: np.ndarray, attention_mask: np.ndarray,
                               labels: Optional[np.ndarray], lang_embeds: np.ndarray, media_token_id: int,
                               pad_token_id: int, num_tokens_per_vis: int, padding_side: str = "right"):
    """``VLMWithLanguageStream._prepare_inputs_for_forward`` (src/vlm.py:445-603) for the no-KV-cache case.

    ``lang_embeds`` = DecoupledEmbedding(lang_x), shape (B, T, d).  Returns dict with
    ``inputs_embeds (B,L,d)``, ``attention_mask (B,1,L,L) int64``, ``labels (B,L) or None`` and, in
    addition to the reference, ``spans``: per-sample list of clamped unlock rectangles and
    ``mask_1d`` (B,L) - what the HIP path consumes instead of the dense mask.
    Multi-image samples raise, exactly like the reference does.
    """
    B = lang_x.shape[0]
    embeds, masks, labs, spans, m1d = [], [], ([] if labels is not None else None), [], []
    for i in range(B):
        img_idxs = np.where(lang_x[i] == media_token_id)[0]
        q = np.where(lang_x[i] == ASSISTANT_TOKEN_ID)[0]
        q = int(q[0]) if len(q) else 0
        if len(img_idxs) == 0:
            embeds.append(lang_embeds[i].copy())
            n = attention_mask[i].shape[0]
            masks.append(make_modality_mutual_mask(attention_mask[i], 0, 0, q))
            spans.append([clamp_span(n, 0, 0, q)])
            m1d.append(attention_mask[i].copy())
            if labels is not None:
                labs.append(labels[i].copy())
            continue
        if len(img_idxs) > 1:
            raise RuntimeError("Tensors must have same number of dimensions: got 3 and 1 "
                               "(reference cannot splice a second image, src/vlm.py:547-554)")
        img = int(img_idxs[0])
        Nv = num_tokens_per_vis
        assert vision_tokens[i][0].shape[0] == Nv
        e = np.concatenate((lang_embeds[i][:img], vision_tokens[i][0], lang_embeds[i][img + 1:]), 0)
        a = np.concatenate((attention_mask[i][:img], np.ones(Nv, dtype=attention_mask.dtype), attention_mask[i][img + 1:]), 0)
        n = a.shape[0]
        masks.append(make_modality_mutual_mask(a, img, img + Nv, q + Nv))
        spans.append([clamp_span(n, img, img + Nv, q + Nv)])
        m1d.append(a)
        embeds.append(e)
        if labels is not None:
            labs.append(np.concatenate((labels[i][:img], np.full(Nv, -100, dtype=labels.dtype), labels[i][img + 1:]), 0))
    out = {
        "inputs_embeds": stack_with_padding(embeds, padding_value=pad_token_id, padding_side=padding_side),
        "attention_mask": stack_with_padding_2d_attention(masks),
        "labels": stack_with_padding(labs, padding_value=-100, padding_side=padding_side) if labels is not None else None,
        "spans": spans,
        "mask_1d": stack_with_padding(m1d, padding_value=0, padding_side="right"),
        "lengths": [e.shape[0] for e in embeds],
    }
    return out


# ----------------------------------------------------------------------------------------------
# a4: Perceiver resampler (the connector)
# ----------------------------------------------------------------------------------------------
def feed_forward(x, ln_w, ln_b, w1, w2):
    """src/helpers.py:32-39: LN -> Linear(d,4d,no bias) -> GELU(erf) -> Linear(4d,d,no bias)."""
    h = layer_norm(x, ln_w, ln_b)
    return linear(gelu_erf(linear(h, w1)), w2)


def perceiver_attention(x, latents, p: Dict[str, np.ndarray], heads=8, dim_head=64):
    """src/helpers.py:76-102.  x (b,T,n1,D), latents (b,T,n2,D)."""
    x = layer_norm(x, p["norm_media.weight"], p["norm_media.bias"])
    latents = layer_norm(latents, p["norm_latents.weight"], p["norm_latents.bias"])
    q = linear(latents, p["to_q.weight"])
    kv = linear(np.concatenate((x, latents), -2), p["to_kv.weight"])
    inner = heads * dim_head
    k, v = kv[..., :inner], kv[..., inner:]

    def split(t):  # b t n (h d) -> b h t n d
        b, T, n, _ = t.shape
        return t.reshape(b, T, n, heads, dim_head).transpose(0, 3, 1, 2, 4)

    q, k, v = split(q), split(k), split(v)
    q = q * np.float32(dim_head ** -0.5)
    sim = q @ np.swapaxes(k, -1, -2)
    sim = sim - sim.max(-1, keepdims=True)
    attn = softmax(sim, -1)
    out = attn @ v
    b, h, T, n, d = out.shape
    out = out.transpose(0, 2, 3, 1, 4).reshape(b, T, n, h * d)
    return linear(out, p["to_out.weight"])


def _sub(p: Dict[str, np.ndarray], prefix: str) -> Dict[str, np.ndarray]:
    return {k[len(prefix):]: v for k, v in p.items() if k.startswith(prefix)}


def perceiver_resampler(x, p: Dict[str, np.ndarray], depth=6, heads=8, dim_head=64):
    """src/helpers.py:170-199.  x (b,T,F,v,D) -> (b,T,n,dim_inner).  No frame/media-time embeddings
    (AKI constructs it without max_num_media/max_num_frames, src/aki.py:40-43)."""
    b, T, F, v, D = x.shape
    x = x.reshape(b, T, F * v, D)
    lat = np.broadcast_to(p["latents"], (b, T) + p["latents"].shape).astype(x.dtype)
    for l in range(depth):
        lat = perceiver_attention(x, lat, _sub(p, f"layers.{l}.0."), heads, dim_head) + lat
        lat = feed_forward(lat, p[f"layers.{l}.1.0.weight"], p[f"layers.{l}.1.0.bias"],
                           p[f"layers.{l}.1.1.weight"], p[f"layers.{l}.1.3.weight"]) + lat
    out = layer_norm(lat, p["norm.weight"], p["norm.bias"])
    if "projection.weight" in p:
        out = linear(out, p["projection.weight"], p["projection.bias"])
    return out


# ----------------------------------------------------------------------------------------------
# a9/a10/a11: Phi-3 decoder with the 4.41.2 mask hand-off
# ----------------------------------------------------------------------------------------------
def invert_mask_441(mask01: np.ndarray, dtype_min: float = F32_MIN) -> np.ndarray:
    """transformers==4.41.2 ``_prepare_4d_causal_attention_mask`` for a 4-D 0/1 input:
    ``inverted = 1 - mask; inverted.masked_fill(inverted.bool(), finfo(dtype).min)``."""
    inv = 1.0 - mask01.astype(np.float32)
    return np.where(inv.astype(bool), np.float32(dtype_min), inv).astype(np.float32)


def rope_inv_freq(head_dim: int, theta: float = 10000.0, ext_factors: Optional[np.ndarray] = None) -> np.ndarray:
    """HF:phi3/modeling_phi3.py:85-106 (default) and modeling_rope_utils longrope
    (inv_freq = 1 / (ext_factors * theta^(2i/d)))."""
    base = theta ** (np.arange(0, head_dim, 2, dtype=np.float32) / np.float32(head_dim))
    if ext_factors is not None:
        base = np.asarray(ext_factors, dtype=np.float32) * base
    return (1.0 / base).astype(np.float32)


def rope_cos_sin(position_ids: np.ndarray, head_dim: int, theta: float = 10000.0,
                 ext_factors: Optional[np.ndarray] = None, attention_scaling: float = 1.0):
    """HF:phi3/modeling_phi3.py:108-124: freqs = pos x inv_freq (fp32), emb = cat(freqs, freqs)."""
    inv = rope_inv_freq(head_dim, theta, ext_factors)
    freqs = position_ids.astype(np.float32)[..., None] * inv[None, :]
    emb = np.concatenate((freqs, freqs), -1)
    return (np.cos(emb) * np.float32(attention_scaling)).astype(np.float32), \
           (np.sin(emb) * np.float32(attention_scaling)).astype(np.float32)


def longrope_attention_scaling(max_pos: int, orig_max_pos: int) -> float:
    f = max_pos / orig_max_pos
    return 1.0 if f <= 1.0 else math.sqrt(1 + math.log(f) / math.log(orig_max_pos))


def rotate_half(x):
    h = x.shape[-1] // 2
    return np.concatenate((-x[..., h:], x[..., :h]), -1)


def apply_rope(q, k, cos, sin):
    """HF:phi3/modeling_phi3.py:170-197 with unsqueeze_dim=1; q,k (B,H,L,D), cos/sin (B or 1, L, D)."""
    cos = cos[:, None]
    sin = sin[:, None]
    return q * cos + rotate_half(q) * sin, k * cos + rotate_half(k) * sin


def mma_attention_core(q, k, v, add_mask, scaling, emulate_bf16=False):
    """eager_attention_forward, HF:phi3/modeling_phi3.py:145-167 (MHA: n_rep = 1).
    q,k,v (B,H,L,D); add_mask (B,1,L,L) additive.  Returns (B,L,H*D)."""
    r = bf16_round if emulate_bf16 else (lambda t: t)
    s = r(r(q @ np.swapaxes(k, -1, -2)) * np.float32(scaling))
    s = r(s + add_mask)
    p = r(softmax(s.astype(np.float32), -1))
    o = r(p @ v)
    B, H, L, D = o.shape
    return o.transpose(0, 2, 1, 3).reshape(B, L, H * D)


def mma_attention_core_spans(q, k, v, mask_1d, spans, scaling):
    """Span-driven restatement of the same arithmetic (what the HIP kernel implements): no dense
    mask; rows with no visible column get the uniform softmax that the finfo.min hand-off yields
    (SURVEY.md section 3.3).  fp32.  spans[b] = list of (row_lo,row_hi,col_lo,col_hi)."""
    B, H, L, D = q.shape
    out = np.zeros((B, L, H * D), dtype=np.float32)
    r = np.arange(L)[:, None]
    c = np.arange(L)[None, :]
    for b in range(B):
        vis = c <= r
        for (r0, r1, c0, c1) in spans[b]:
            vis = vis | ((r >= r0) & (r < r1) & (c >= c0) & (c < c1))
        vis = vis & mask_1d[b].astype(bool)[None, :]
        dead = ~vis.any(-1)
        for h in range(H):
            s = (q[b, h] @ k[b, h].T) * np.float32(scaling)
            s = np.where(vis, s, -np.inf)
            s[dead] = 0.0
            p = softmax(s.astype(np.float32), -1)
            out[b, :, h * D:(h + 1) * D] = p @ v[b, h]
    return out


def phi3_attention(x, w_qkv, w_o, cos, sin, add_mask, n_heads, emulate_bf16=False):
    """Phi3Attention.forward, HF:phi3/modeling_phi3.py:218-263 (no KV cache)."""
    r = bf16_round if emulate_bf16 else (lambda t: t)
    B, L, d = x.shape
    Dh = d // n_heads
    qkv = r(linear(x, w_qkv))
    q, k, v = qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:]

    def heads(t):
        return t.reshape(B, L, n_heads, Dh).transpose(0, 2, 1, 3)

    q, k, v = heads(q), heads(k), heads(v)
    if emulate_bf16:
        cos, sin = bf16_round(cos), bf16_round(sin)
        cu, su = cos[:, None], sin[:, None]
        q = r(r(q * cu) + r(rotate_half(q) * su))
        k = r(r(k * cu) + r(rotate_half(k) * su))
    else:
        q, k = apply_rope(q, k, cos, sin)
    o = mma_attention_core(q, k, v, add_mask, Dh ** -0.5, emulate_bf16)
    return r(linear(o, w_o))


def phi3_mlp(x, w_gate_up, w_down):
    """HF:phi3/modeling_phi3.py:49-64."""
    up = linear(x, w_gate_up)
    half = up.shape[-1] // 2
    gate, u = up[..., :half], up[..., half:]
    return linear(u * silu(gate), w_down)


def phi3_decoder_layer(h, p: Dict[str, np.ndarray], cos, sin, add_mask, n_heads, eps=1e-5):
    """HF:phi3/modeling_phi3.py:287-328."""
    x = rms_norm(h, p["input_layernorm.weight"], eps)
    h = h + phi3_attention(x, p["self_attn.qkv_proj.weight"], p["self_attn.o_proj.weight"], cos, sin, add_mask, n_heads)
    x = rms_norm(h, p["post_attention_layernorm.weight"], eps)
    return h + phi3_mlp(x, p["mlp.gate_up_proj.weight"], p["mlp.down_proj.weight"])


def phi3_lm_forward(inputs_embeds, mask01_4d, p: Dict[str, np.ndarray], n_layers, n_heads, max_original_id,
                    theta=10000.0, eps=1e-5, position_ids=None):
    """Phi3ForCausalLM on inputs_embeds with a 4-D 0/1 mask under 4.41.2 semantics; ``p`` uses the
    reference's state-dict names below ``lang_model.``.  Returns logits (B,L,V')."""
    B, L, d = inputs_embeds.shape
    pos = np.arange(L)[None] if position_ids is None else position_ids
    cos, sin = rope_cos_sin(pos, d // n_heads, theta)
    add = invert_mask_441(mask01_4d)
    h = inputs_embeds.astype(np.float32)
    for l in range(n_layers):
        h = phi3_decoder_layer(h, _sub(p, f"model.layers.{l}."), cos, sin, add, n_heads, eps)
    h = rms_norm(h, p["model.norm.weight"], eps)
    return decoupled_linear(h, p["lm_head.weight"], p.get("lm_head.bias"), p.get("lm_head.additional_fc.weight"),
                            p.get("lm_head.additional_fc.bias"), max_original_id)


def causal_lm_loss(logits, labels, ignore_index=-100):
    """HF ForCausalLMLoss: shift by one, mean CE over non-ignored targets, fp32."""
    lg = logits[:, :-1].astype(np.float32).reshape(-1, logits.shape[-1])
    tg = labels[:, 1:].reshape(-1)
    keep = tg != ignore_index
    lg, tg = lg[keep], tg[keep]
    m = lg.max(-1, keepdims=True)
    lse = (m + np.log(np.exp(lg - m).sum(-1, keepdims=True)))[:, 0]
    return float((lse - lg[np.arange(len(tg)), tg]).mean())


# ----------------------------------------------------------------------------------------------
# a2/a3: SigLIP vision tower
# ----------------------------------------------------------------------------------------------
def siglip_patch_embed(pixels, w, b, pos_emb):
    """SiglipVisionEmbeddings.forward, HF:siglip/modeling_siglip.py:175-185.
    pixels (N,3,S,S), w (E,3,P,P), b (E,), pos_emb (G*G,E) -> (N,G*G,E)."""
    N, C, S, _ = pixels.shape
    E, _, P, _ = w.shape
    G = S // P
    x = pixels[:, :, :G * P, :G * P].reshape(N, C, G, P, G, P).transpose(0, 2, 4, 1, 3, 5).reshape(N, G * G, C * P * P)
    return (x @ w.reshape(E, -1).T + b + pos_emb[None]).astype(pixels.dtype)


def siglip_encoder_layer(h, p: Dict[str, np.ndarray], n_heads, eps=1e-6):
    """HF:siglip/modeling_siglip.py:325-357 (+ 250-307 attention, 310-322 MLP)."""
    N, L, E = h.shape
    Dh = E // n_heads
    x = layer_norm(h, p["layer_norm1.weight"], p["layer_norm1.bias"], eps)

    def heads(t):
        return t.reshape(N, L, n_heads, Dh).transpose(0, 2, 1, 3)

    q = heads(linear(x, p["self_attn.q_proj.weight"], p["self_attn.q_proj.bias"]))
    k = heads(linear(x, p["self_attn.k_proj.weight"], p["self_attn.k_proj.bias"]))
    v = heads(linear(x, p["self_attn.v_proj.weight"], p["self_attn.v_proj.bias"]))
    a = softmax((q @ np.swapaxes(k, -1, -2)) * np.float32(Dh ** -0.5), -1) @ v
    a = a.transpose(0, 2, 1, 3).reshape(N, L, E)
    h = h + linear(a, p["self_attn.out_proj.weight"], p["self_attn.out_proj.bias"])
    x = layer_norm(h, p["layer_norm2.weight"], p["layer_norm2.bias"], eps)
    x = linear(gelu_tanh(linear(x, p["mlp.fc1.weight"], p["mlp.fc1.bias"])), p["mlp.fc2.weight"], p["mlp.fc2.bias"])
    return h + x


def siglip_vision_forward(pixels, p: Dict[str, np.ndarray], n_layers, n_heads, eps=1e-6):
    """SiglipVisionModel(...).last_hidden_state, HF:siglip/modeling_siglip.py:576-620 (head unused by AKI)."""
    h = siglip_patch_embed(pixels, p["embeddings.patch_embedding.weight"], p["embeddings.patch_embedding.bias"],
                           p["embeddings.position_embedding.weight"])
    for l in range(n_layers):
        h = siglip_encoder_layer(h, _sub(p, f"encoder.layers.{l}."), n_heads, eps)
    return layer_norm(h, p["post_layernorm.weight"], p["post_layernorm.bias"], eps)


# ----------------------------------------------------------------------------------------------
# a1: AKI.forward
# ----------------------------------------------------------------------------------------------
def aki_forward(p: Dict[str, np.ndarray], cfg: Dict, vision_x, lang_x, attention_mask, labels=None):
    """``AKI.forward`` (src/aki.py:65-134) end to end in fp32.  ``p`` is the full state dict
    (reference key names); cfg holds the handful of structural integers."""
    vt = None
    if vision_x is not None:
        b, T, F = vision_x.shape[:3]
        px = vision_x.reshape((b * T * F,) + vision_x.shape[3:])
        feats = siglip_vision_forward(px, _sub(p, "vision_encoder."), cfg["vis_layers"], cfg["vis_heads"])
        feats = feats.reshape(b, T, F, feats.shape[1], feats.shape[2])            # src/vlm.py:206
        vt = perceiver_resampler(feats, _sub(p, "vision_tokenizer."), cfg.get("perc_depth", 6),
                                 cfg.get("perc_heads", 8), cfg.get("perc_dim_head", 64))
    lm = _sub(p, "lang_model.")
    emb = decoupled_embedding(lang_x, lm["model.embed_tokens.weight"],
                              lm.get("model.embed_tokens.additional_embedding.weight"), cfg["max_original_id"])
    if vt is None:
        raise NotImplementedError("text-only path goes through input_ids in the reference (src/vlm.py:471-476)")
    prep = prepare_inputs_for_forward(vt, lang_x, attention_mask, labels, emb, cfg["media_token_id"],
                                      cfg["pad_token_id"], cfg["num_vision_tokens"], "right")
    logits = phi3_lm_forward(prep["inputs_embeds"], prep["attention_mask"], lm, cfg["lm_layers"], cfg["lm_heads"],
                             cfg["max_original_id"], cfg.get("rope_theta", 10000.0), cfg.get("rms_eps", 1e-5))
    loss = causal_lm_loss(logits, prep["labels"]) if labels is not None else None
    return {"logits": logits, "loss": loss, "prep": prep}
