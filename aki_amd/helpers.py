"""Connector + embedding helpers: the host-side mirror of src/helpers.py (same class names, constructor
arguments, parameter names and therefore state-dict keys), with the arithmetic done by the HIP kernels.

Reference: /root/reference/codes/open_flamingo/src/helpers.py
  FeedForward :32-39 | PerceiverAttention :62-102 | PerceiverResampler :105-199
  DecoupledEmbedding :350-492 | DecoupledLinear :495-613 | VLMOutputWithPast :16-25
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import torch
import torch.nn.functional as F
from torch import nn

from . import ops
from . import train_ops as TR

try:  # the HF output container the reference returns (src/helpers.py:11,16)
    from transformers.modeling_outputs import CausalLMOutputWithPast
except Exception:  # pragma: no cover - transformers is present in this image
    @dataclass
    class CausalLMOutputWithPast:  # type: ignore
        loss: Optional[torch.Tensor] = None
        logits: Optional[torch.Tensor] = None
        past_key_values: Optional[object] = None
        hidden_states: Optional[object] = None
        attentions: Optional[object] = None

        def __getitem__(self, i):
            return [v for v in (self.loss, self.logits, self.past_key_values) if v is not None][i]


@dataclass
class VLMOutputWithPast(CausalLMOutputWithPast):
    past_media_locations: Optional[torch.Tensor] = None
    past_vision_tokens: Optional[torch.Tensor] = None


def _ag(x, *params) -> bool:
    return (torch.is_grad_enabled() and x.dtype == torch.bfloat16
            and (x.requires_grad or any(p is not None and p.requires_grad for p in params)))


def exists(val):
    return val is not None


def FeedForward(dim, mult=4):
    """Same container layout as the reference (indices 0,1,3 carry parameters); the forward pass of the
    block is executed by ``ops.connector_mlp`` in :class:`PerceiverResampler`."""
    inner_dim = int(dim * mult)
    return nn.Sequential(
        nn.LayerNorm(dim),
        nn.Linear(dim, inner_dim, bias=False),
        nn.GELU(),
        nn.Linear(inner_dim, dim, bias=False),
    )


class VisionTokenizer(nn.Module):
    def __init__(self, dim_media, num_tokens_per_media):
        super().__init__()
        self.dim_media = dim_media
        self.num_tokens_per_media = num_tokens_per_media


class PerceiverAttention(nn.Module):
    def __init__(self, *, dim, dim_head=64, heads=8):
        super().__init__()
        self.scale = dim_head ** -0.5
        self.heads = heads
        self.dim_head = dim_head
        inner_dim = dim_head * heads
        self.norm_media = nn.LayerNorm(dim)
        self.norm_latents = nn.LayerNorm(dim)
        self.to_q = nn.Linear(dim, inner_dim, bias=False)
        self.to_kv = nn.Linear(dim, inner_dim * 2, bias=False)
        self.to_out = nn.Linear(inner_dim, dim, bias=False)

    def forward(self, x, latents):
        """x (b,T,n1,D), latents (b,T,n2,D) -> to_out(attn) + latents  (the residual of src/helpers.py:193 is
        fused into the output projection's epilogue)."""
        b, T_, n1, D = x.shape
        n2 = latents.shape[2]
        h, dh = self.heads, self.dim_head
        if _ag(latents, *self.parameters()):       # training: same kernels, outputs kept for the HIP backward
            xn = TR.layernorm(x, self.norm_media.weight, self.norm_media.bias, self.norm_media.eps)
            ln = TR.layernorm(latents, self.norm_latents.weight, self.norm_latents.bias, self.norm_latents.eps)
            q = TR.linear(ln, self.to_q.weight)
            kv = TR.linear(torch.cat((xn, ln), dim=-2), self.to_kv.weight).view(b * T_, n1 + n2, 2, h, dh)
            out = TR.PlainAttnFn.apply(q.view(b * T_, n2, h, dh), kv[:, :, 0], kv[:, :, 1], self.scale)
            return TR.linear(out.view(b, T_, n2, h * dh), self.to_out.weight, None, latents)
        xn = ops.layernorm(x, self.norm_media.weight, self.norm_media.bias, self.norm_media.eps)
        ln = ops.layernorm(latents, self.norm_latents.weight, self.norm_latents.bias, self.norm_latents.eps)
        q = ops.linear(ln, self.to_q.weight)
        kv = ops.linear(torch.cat((xn, ln), dim=-2), self.to_kv.weight)
        kv = kv.view(b * T_, n1 + n2, 2, h, dh)
        # 8 x 64 heads, 144 queries over 873 keys: HIP attention reading k / v in place from the fused kv projection
        # (q * scale then softmax(sim - max) of the reference == softmax(scale * q k^T))
        out = ops.attention(q.view(b * T_, n2, h, dh), kv[:, :, 0], kv[:, :, 1], self.scale)
        return ops.linear(out.view(b, T_, n2, h * dh), self.to_out.weight, residual=latents)


class PerceiverResampler(VisionTokenizer):
    def __init__(self, *, dim, dim_inner=None, depth=6, dim_head=64, heads=8, num_latents=64, max_num_media=None,
                 max_num_frames=None, ff_mult=4):
        if dim_inner is not None:
            projection = nn.Linear(dim, dim_inner)
        else:
            projection = None
            dim_inner = dim
        super().__init__(dim_media=dim, num_tokens_per_media=num_latents)
        self.projection = projection
        self.latents = nn.Parameter(torch.randn(num_latents, dim))
        self.frame_embs = nn.Parameter(torch.randn(max_num_frames, dim)) if exists(max_num_frames) else None
        self.media_time_embs = nn.Parameter(torch.randn(max_num_media, 1, dim)) if exists(max_num_media) else None
        self.layers = nn.ModuleList([])
        for _ in range(depth):
            self.layers.append(nn.ModuleList([PerceiverAttention(dim=dim, dim_head=dim_head, heads=heads),
                                              FeedForward(dim=dim, mult=ff_mult)]))
        self.norm = nn.LayerNorm(dim)

    use_layer_stack = True      # one (sample, image) pair: the ~55 launches of the 6 layers issued by ONE C call (csrc/stack.hip: aki_perceiver_stack_fwd)

    def _can_stack(self, x) -> bool:
        if not (self.use_layer_stack and ops.stack_enabled() and x.is_cuda and x.dtype == torch.bfloat16 and x.is_contiguous()):
            return False
        eps = self.norm.eps
        for attn, ff in self.layers:
            if (attn.norm_media.eps != eps or attn.norm_latents.eps != eps or ff[0].eps != eps
                    or attn.heads != self.layers[0][0].heads or attn.dim_head != self.layers[0][0].dim_head):
                return False
        return not ops.python_must_run_between(self.layers)

    def _forward_stack(self, x, latents):
        """= the inference loop below for one (sample, image) pair, issued by aki_perceiver_stack_fwd: same launches, same arguments."""
        tb = getattr(self, "_stack_table", None)
        if tb is None:
            tb = self._stack_table = ops.LayerTable(ops.L.PerceiverLayer)
        # the LIVE parameter objects on every call (load_state_dict(assign=True) and attribute assignment replace objects: a cached list
        # would go on signing the old ones); object identity is part of the signature
        live = []
        for attn, ff in self.layers:
            live += [attn.norm_media.weight, attn.norm_media.bias, attn.norm_latents.weight, attn.norm_latents.bias, attn.to_q.weight,
                     attn.to_kv.weight, attn.to_out.weight, ff[0].weight, ff[0].bias, ff[1].weight, ff[3].weight]
        sig = (tuple(map(id, live)), ops.params_signature(live))
        if tb.sig != sig:
            rows = []
            for attn, ff in self.layers:
                rows.append((attn.norm_media.weight, attn.norm_media.bias, attn.norm_latents.weight, attn.norm_latents.bias, attn.to_q.weight,
                             attn.to_kv.weight, attn.to_out.weight, ff[0].weight, ff[0].bias, ff[1].weight, ff[3].weight))
            for r in rows:
                for t_ in r:
                    if t_ is not None and not t_.is_contiguous():
                        raise ops.AkiError("perceiver stack: weights must be contiguous")
            tb.get(rows)
            tb.sig = sig
        a0 = self.layers[0][0]
        proj = self.projection
        return ops.perceiver_stack(tb.arr, len(self.layers), x.reshape(-1, x.shape[-1]), latents.reshape(-1, latents.shape[-1]), self.norm.weight, self.norm.bias,
                                   None if proj is None else proj.weight, None if proj is None else proj.bias, a0.heads, a0.dim_head,
                                   self.layers[0][1][1].weight.shape[0], a0.scale, self.norm.eps)

    def forward(self, x):
        """x (b,T,F,v,D) -> (b,T,n,dim_inner)   (src/helpers.py:170-199)."""
        b, T, Fr, v = x.shape[:4]
        if exists(self.frame_embs):
            x = x + self.frame_embs[:Fr][None, None, :, None, :]
        x = x.reshape(b, T, Fr * v, x.shape[-1])
        if exists(self.media_time_embs):
            x = x + self.media_time_embs[:T]
        latents = self.latents.to(x.dtype)[None, None].expand(b, T, -1, -1).contiguous()
        if _ag(latents, *self.parameters()):
            for attn, ff in self.layers:
                latents = attn(x, latents)
                hmid = TR.GeluFn.apply(TR.linear(TR.layernorm(latents, ff[0].weight, ff[0].bias, ff[0].eps), ff[1].weight))
                latents = TR.linear(hmid, ff[3].weight, None, latents)
            latents = TR.layernorm(latents, self.norm.weight, self.norm.bias, self.norm.eps)
            if exists(self.projection):
                return TR.linear(latents, self.projection.weight, self.projection.bias, None)
            return latents
        if b * T == 1 and self._can_stack(x):
            return self._forward_stack(x, latents).view(b, T, latents.shape[2], -1)
        for attn, ff in self.layers:
            latents = attn(x, latents)                                     # attention + residual
            latents = ops.connector_mlp(latents, ff[0].weight, ff[0].bias, ff[1].weight, ff[3].weight, ff[0].eps)
        if exists(self.projection):
            return ops.connector_proj(latents, self.norm.weight, self.norm.bias, self.projection.weight,
                                      self.projection.bias, self.norm.eps)
        return ops.layernorm(latents, self.norm.weight, self.norm.bias, self.norm.eps)


def _table_dims(given: Optional[torch.Tensor], declared_rows, declared_cols, what: str):
    """(rows, cols) of a parameter table that is either handed over ready-made or described by its two sizes."""
    if given is None:
        if declared_rows is None or declared_cols is None:
            raise AssertionError(f"{what}: without a ready-made tensor both sizes have to be given")
        return int(declared_rows), int(declared_cols)
    rows, cols = given.shape
    for declared, actual in ((declared_rows, rows), (declared_cols, cols)):
        if declared is not None and declared != actual:
            raise AssertionError(f"{what}: the tensor is {rows} x {cols}, which contradicts the declared size {declared}")
    return int(rows), int(cols)


class DecoupledEmbedding(nn.Embedding):
    """Embedding table split into the tokenizer's original rows (`weight`, frozen by default) and the rows of the special
    tokens added for AKI (`additional_embedding.weight`, always trainable) - parameter names and constructor arguments of
    src/helpers.py:350-492, because checkpoints and `VLM.__init__` depend on them.  On the product path the lookup happens
    inside the splice kernel (aki_splice_fwd reads both tables); `forward` is the standalone form of the same gather."""

    def __init__(self, max_original_id: int, num_additional_embeddings: int = 0, _weight: torch.Tensor = None,
                 num_original_embeddings: int = None, embedding_dim: int = None, partially_freeze=True, device=None,
                 dtype=None, pad_token_id=None) -> None:
        if pad_token_id is not None and pad_token_id > max_original_id:
            raise ValueError(f"pad_token_id={pad_token_id} lies beyond max_original_id={max_original_id}: the padding row has "
                             "to be one of the original rows (pass pad_token_id=None for a tokenizer without a pad token)")
        rows, width = _table_dims(_weight, num_original_embeddings, embedding_dim, "DecoupledEmbedding")
        super().__init__(rows, width, padding_idx=pad_token_id, _weight=_weight, device=device, dtype=dtype)
        self.max_original_id = max_original_id
        self.padding_idx = pad_token_id
        self.num_additional_embeddings = num_additional_embeddings
        if num_additional_embeddings > 0:
            self.additional_embedding = nn.Embedding(num_additional_embeddings, width, device=device, dtype=dtype)
        self.set_requires_grad(require_regular_grad=not partially_freeze, require_additional_grad=True)

    def set_requires_grad(self, require_regular_grad, require_additional_grad):
        self.weight.requires_grad_(require_regular_grad)
        self.additional_embedding.requires_grad_(require_additional_grad)

    def forward(self, input_ids):
        if self.num_additional_embeddings == 0:
            return F.embedding(input_ids, self.weight)
        hi = input_ids > self.max_original_id
        low = torch.where(hi, torch.zeros_like(input_ids), input_ids)
        full = F.embedding(low, self.weight)
        add = self.additional_embedding(torch.where(hi, input_ids - self.max_original_id - 1, torch.zeros_like(input_ids)))
        return torch.where(hi[..., None], add, full)

    def extra_repr(self) -> str:
        frozen = "frozen" if not self.weight.requires_grad else "trainable"
        return (f"{self.max_original_id + 1} original rows ({frozen}) + {self.num_additional_embeddings} additional rows, "
                f"width {self.embedding_dim}")


class DecoupledLinear(nn.Linear):
    """Output head split the same way (src/helpers.py:495-613): logits = (x W^T)[..., :max_original_id+1] ++ x W_add^T with
    `weight` / `bias` for the original vocabulary and `additional_fc` for the added tokens.  Executed as ONE HIP GEMM over
    the row-concatenated weight, which is rebuilt only when a parameter changes."""

    def __init__(self, max_original_id: int, additional_out_features: int = 0, _weight: torch.Tensor = None,
                 _bias: torch.Tensor = None, in_features: int = None, original_out_features: int = None, bias: bool = True,
                 partially_freeze: bool = True, device=None, dtype=None) -> None:
        n_out, n_in = _table_dims(_weight, original_out_features, in_features, "DecoupledLinear")
        if _bias is not None and not bias:
            raise AssertionError("DecoupledLinear: a bias tensor was handed over although bias=False")
        super().__init__(n_in, n_out, bias, device, dtype)
        if _weight is not None:
            self.weight = nn.Parameter(_weight)
        if _bias is not None:
            self.bias = nn.Parameter(_bias)
        self.in_features, self.original_out_features = n_in, n_out
        self.max_original_id = max_original_id
        self.additional_out_features = additional_out_features
        self.has_bias = bias
        if additional_out_features > 0:
            self.additional_fc = nn.Linear(n_in, additional_out_features, bias=bias, device=device, dtype=dtype)
        self.set_requires_grad(require_regular_grad=not partially_freeze, require_additional_grad=True)
        self._fused = None

    def set_requires_grad(self, require_regular_grad, require_additional_grad):
        for p in (self.weight, self.bias if self.has_bias else None):
            if p is not None:
                p.requires_grad_(require_regular_grad)
        self.additional_fc.requires_grad_(require_additional_grad)

    def _fused_weight(self):
        n0 = self.max_original_id + 1
        extra = self.additional_out_features
        key = (TR._EPOCH, self.weight.data_ptr(), self.weight._version, self.weight.dtype, self.weight.device,
               self.additional_fc.weight._version if extra else 0, self.additional_fc.weight.data_ptr() if extra else 0)
        if self._fused is None or self._fused[0] != key:
            n = n0 + extra
            npad = (n + 3) // 4 * 4
            w = torch.zeros((npad, self.in_features), dtype=self.weight.dtype, device=self.weight.device)
            w[:n0] = self.weight.detach()[:n0]
            b = None
            if self.has_bias and self.bias is not None:
                b = torch.zeros((npad,), dtype=self.weight.dtype, device=self.weight.device)
                b[:n0] = self.bias.detach()[:n0]
            if extra:
                w[n0:n] = self.additional_fc.weight.detach()
                if b is not None and self.additional_fc.bias is not None:
                    b[n0:n] = self.additional_fc.bias.detach()
            self._fused = (key, w, b, n)
        return self._fused[1], self._fused[2], self._fused[3]

    def forward(self, input: torch.Tensor) -> torch.Tensor:
        w, b, n = self._fused_weight()
        out = ops.linear(input, w, bias=b)
        return out[..., :n]

    def forward_normed(self, input: torch.Tensor, rms_weight: torch.Tensor, eps: float) -> torch.Tensor:
        """forward(rmsnorm(input)) for the rows of a decode step (final norm fused into the weight-streaming head)."""
        w, b, n = self._fused_weight()
        return ops.decode_linear(input, w, rms_weight, eps, bias=b)[..., :n]

    def extra_repr(self) -> str:
        frozen = "frozen" if not self.weight.requires_grad else "trainable"
        return (f"{self.in_features} -> {self.max_original_id + 1} original ({frozen}) + {self.additional_out_features} "
                f"additional outputs, bias={self.bias is not None}")
