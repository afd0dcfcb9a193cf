"""MI355X-native SigLIP vision tower (frozen in AKI).  Same module tree / parameter names as HF
``SiglipVisionModel(...).vision_model`` (HF:siglip/modeling_siglip.py:116-185,250-357,553-620), class name
``SiglipVisionTransformer`` as the reference dispatches on it (src/vlm.py:9,202).

Patch embedding = the HIP patch-embed kernel (im2col + MFMA GEMM with bias and position-embedding epilogue).
Encoder layers: LayerNorm, fused QKV / out / MLP projections on the HIP GEMM (bias, GELU-tanh and residual
epilogues) and the 16 x 72 attention on the HIP non-causal attention kernel (aki_attn_fwd), which reads q/k/v in
place from the fused QKV output.
"""
from __future__ import annotations

from types import SimpleNamespace

import torch
import torch.nn.functional as F
from torch import nn

from . import ops


def make_siglip_config(**kw):
    d = dict(hidden_size=1152, intermediate_size=4304, num_hidden_layers=27, num_attention_heads=16, num_channels=3,
             image_size=384, patch_size=14, layer_norm_eps=1e-6, hidden_act="gelu_pytorch_tanh")
    d.update(kw)
    return SimpleNamespace(**d)


_Prepared = ops.Prepared


def _epoch(params) -> int:
    """The trainers write weights through raw pointers without bumping tensor versions and announce it through the weight epoch
    (train_ops.bump_weight_epoch): every cached transform of a TRAINABLE weight is keyed on it, the same rule as in phi3.py.  A trainer only
    ever writes parameters with requires_grad - the frozen tower of the reference's recipes keeps its folded copies across optimizer steps
    (keyed on the epoch they were rebuilt in every training step: 27 x 3 transforms, ≈6 ms of small torch kernels and as much idle GPU)."""
    if not any(p.requires_grad for p in params):
        return 0
    from . import train_ops as T
    return T._EPOCH


def fold_layernorm(w: torch.Tensor, b: torch.Tensor, ln: nn.LayerNorm):
    """LayerNorm(x) @ W^T + b = rstd * (x @ W'^T - mean * c) + b'  with  W' = W diag(gamma) (rounded to W's dtype),
    c[n] = sum_k W'[n][k] (f32, of the ROUNDED W' - that is what the MFMA multiplies) and b' = W beta + b."""
    wf = ops.fold_gain(w, ln.weight)
    c = torch.zeros(((w.shape[0] + 3) // 4 * 4,), dtype=torch.float32, device=w.device)
    c[: w.shape[0]] = wf.float().sum(1)
    ct = torch.float64 if w.dtype == torch.float64 else torch.float32
    bf = (w.to(ct) @ ln.bias.detach().to(ct) + (b.to(ct) if b is not None else 0.0)).to(w.dtype).contiguous()
    return wf, bf, c


class SiglipVisionEmbeddings(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = config
        self.embed_dim = config.hidden_size
        self.image_size = config.image_size
        self.patch_size = config.patch_size
        self.patch_embedding = nn.Conv2d(config.num_channels, self.embed_dim, kernel_size=self.patch_size,
                                         stride=self.patch_size, padding="valid")
        self.num_patches = (self.image_size // self.patch_size) ** 2
        self.num_positions = self.num_patches
        self.position_embedding = nn.Embedding(self.num_positions, self.embed_dim)
        self._prep = _Prepared()

    def _pos_for(self, grid: int):
        """Learned position table; bicubic interpolation to another grid (HF:siglip 137-173) for the 336 px
        throughput configuration - parity unpinned by the reference, which only ever runs its native size."""
        w = self.position_embedding.weight
        g0 = int(round(self.num_positions ** 0.5))
        if grid == g0:
            return w
        return self._prep.get(f"pos{grid}", [w], lambda: F.interpolate(
            w.float().reshape(1, g0, g0, -1).permute(0, 3, 1, 2), size=(grid, grid), mode="bicubic", align_corners=False
        ).permute(0, 2, 3, 1).reshape(grid * grid, -1).to(w.dtype).contiguous())

    def forward(self, pixel_values, interpolate_pos_encoding=False):
        w = self.patch_embedding.weight
        wp = self._prep.get("w", [w], lambda: ops.pad_k(w.reshape(w.shape[0], -1)))
        S = pixel_values.shape[-1]
        grid = S // self.patch_size
        if grid * grid != self.num_positions and not interpolate_pos_encoding:
            raise ValueError(f"input size {S} does not match the position table ({self.num_positions} patches); "
                             "pass interpolate_pos_encoding=True")
        return ops.patch_embed(pixel_values.to(w.dtype), wp, self.patch_embedding.bias, self._pos_for(grid), self.patch_size)


class SiglipAttention(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.embed_dim = config.hidden_size
        self.num_heads = config.num_attention_heads
        self.head_dim = self.embed_dim // self.num_heads
        self.scale = self.head_dim ** -0.5
        self.k_proj = nn.Linear(self.embed_dim, self.embed_dim)
        self.v_proj = nn.Linear(self.embed_dim, self.embed_dim)
        self.q_proj = nn.Linear(self.embed_dim, self.embed_dim)
        self.out_proj = nn.Linear(self.embed_dim, self.embed_dim)
        self._prep = _Prepared()

    def forward_folded(self, h, st, ln, stats_out, stats_eps):
        """out_proj(attention(qkv(LayerNorm(h)))) + h without the LayerNorm launch: `st` = (1/std, mean) per token of h; the gain
        sits in the cached qkv weight, the shift in its bias, the mean's share mu[m] * sum_k W'[n][k] comes off in the epilogue.
        The out-projection leaves the statistics of its own output in `stats_out` for the next LayerNorm."""
        N, L, E = h.shape
        ps = [self.q_proj.weight, self.k_proj.weight, self.v_proj.weight, self.q_proj.bias, self.k_proj.bias, self.v_proj.bias,
              ln.weight, ln.bias]
        wqkv, bqkv, cqkv = self._prep.get("qkv_ln", ps, lambda: fold_layernorm(torch.cat([p.detach() for p in ps[:3]], 0),
                                                                                torch.cat([p.detach() for p in ps[3:6]], 0), ln), _epoch(ps))
        qkv = ops.linear(h, wqkv, bias=bqkv, row_scale=st.rstd, row_shift=st.mean, col_shift=cqkv)
        qkv = qkv.view(N, L, 3, self.num_heads, self.head_dim)
        a = ops.attention(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], self.scale)
        return ops.linear(a, self.out_proj.weight, bias=self.out_proj.bias, residual=h, stats_out=stats_out, stats_eps=stats_eps)

    def forward(self, x, residual):
        N, L, E = x.shape
        ps = [self.q_proj.weight, self.k_proj.weight, self.v_proj.weight, self.q_proj.bias, self.k_proj.bias, self.v_proj.bias]
        wqkv, bqkv = self._prep.get("qkv", ps, lambda: (torch.cat([p.detach() for p in ps[:3]], 0).contiguous(),
                                                        torch.cat([p.detach() for p in ps[3:]], 0).contiguous()), _epoch(ps))
        qkv = ops.linear(x, wqkv, bias=bqkv).view(N, L, 3, self.num_heads, self.head_dim)
        a = ops.attention(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], self.scale)   # strided views, no copies
        return ops.linear(a, self.out_proj.weight, bias=self.out_proj.bias, residual=residual)


_HBUF = {}      # (leading shape, padded width, dtype, device, stream) -> K-padded fc1 output buffer shared by all layers


def _fc1_buffer(lead, Kp, x):
    key = (tuple(lead), Kp, x.dtype, x.device, torch.cuda.current_stream().cuda_stream)
    hbuf = _HBUF.get(key)
    if hbuf is None:
        if len(_HBUF) > 8:
            _HBUF.clear()
        hbuf = _HBUF[key] = torch.zeros((*lead, Kp), dtype=x.dtype, device=x.device)
    return hbuf


class SiglipMLP(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.fc1 = nn.Linear(config.hidden_size, config.intermediate_size)
        self.fc2 = nn.Linear(config.intermediate_size, config.hidden_size)
        self.act = ops.ACT_GELU_TANH if "tanh" in config.hidden_act else ops.ACT_GELU_ERF
        self._prep = _Prepared()

    def forward(self, x, residual, st=None, ln=None, stats_out=None, stats_eps=0.0):
        """fc2(act(fc1(x))) + residual.  With `st`/`ln`: x is the raw stream and LayerNorm `ln` is folded into fc1 (see
        SiglipAttention.forward_folded); `stats_out` receives the statistics of the result."""
        inter = self.fc1.weight.shape[0]
        Kp = (inter + 63) // 64 * 64
        w2 = self._prep.get("w2", [self.fc2.weight], lambda: ops.pad_k(self.fc2.weight.detach()), _epoch([self.fc2.weight]))
        lead = x.shape[:-1]
        # fc1 writes into a K-padded buffer; the pad columns must be finite zeros for fc2 (zero weights there).  The buffer
        # is kept per shape: fc1 overwrites columns [:inter] every call and nothing ever writes the pad columns, so they are
        # zeroed once instead of a 40 MB fill per layer and step (stream-ordered reuse: the next fc1 runs after this fc2).
        hbuf = _fc1_buffer(lead, Kp, x)
        if st is None:
            ops.linear(x, self.fc1.weight, bias=self.fc1.bias, act=self.act, out=hbuf[..., :inter])
        else:
            w1, b1, c1 = self._prep.get("fc1_ln", [self.fc1.weight, self.fc1.bias, ln.weight, ln.bias],
                                        lambda: fold_layernorm(self.fc1.weight.detach(), self.fc1.bias.detach(), ln),
                                        _epoch([self.fc1.weight, self.fc1.bias, ln.weight, ln.bias]))
            ops.linear(x, w1, bias=b1, act=self.act, out=hbuf[..., :inter], row_scale=st.rstd, row_shift=st.mean, col_shift=c1)
        return ops.linear(hbuf, w2, bias=self.fc2.bias, residual=residual, stats_out=stats_out, stats_eps=stats_eps)


class SiglipEncoderLayer(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.layer_norm1 = nn.LayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.self_attn = SiglipAttention(config)
        self.layer_norm2 = nn.LayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.mlp = SiglipMLP(config)

    def forward_folded(self, h, st, want_stats=True):
        """Inference layer in 5 launches (qkv, attention, out, fc1, fc2): both LayerNorms ride on the GEMMs around them.
        Returns (h_out, statistics of h_out for the next layer's layer_norm1 - every layer shares config.layer_norm_eps)."""
        M = h.numel() // h.shape[-1]
        st2 = ops.new_stats(M, h.device, ln=True)
        h = self.self_attn.forward_folded(h, st, self.layer_norm1, st2, self.layer_norm2.eps)
        st3 = ops.new_stats(M, h.device, ln=True) if want_stats else None
        return self.mlp(h, h, st=st2, ln=self.layer_norm2, stats_out=st3, stats_eps=self.layer_norm1.eps), st3

    def forward(self, h, stats=None, want_stats=True):
        """-> h_out; with `stats` (ops.RowStats of h) -> forward_folded's (h_out, stats of h_out)."""
        if stats is not None:
            return self.forward_folded(h, stats, want_stats)
        h = self.self_attn(ops.layernorm(h, self.layer_norm1.weight, self.layer_norm1.bias, self.layer_norm1.eps), h)
        return self.mlp(ops.layernorm(h, self.layer_norm2.weight, self.layer_norm2.bias, self.layer_norm2.eps), h)


class SiglipEncoder(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.layers = nn.ModuleList([SiglipEncoderLayer(config) for _ in range(config.num_hidden_layers)])

    fold_norms = True      # bf16 inference: LayerNorms folded into the neighbouring GEMMs

    def forward(self, h):
        if self.fold_norms and h.dtype == torch.bfloat16 and not (torch.is_grad_enabled() and (
                h.requires_grad or any(p.requires_grad for p in self.parameters()))):
            if self._can_stack(h):
                return self._forward_stack(h)
            st = ops.row_stats(h, self.layers[0].layer_norm1.eps, ln=True)      # the embeddings' statistics: the one extra pass
            for i, layer in enumerate(self.layers):
                h, st = layer(h, stats=st, want_stats=i + 1 < len(self.layers))
            return h
        for layer in self.layers:
            h = layer(h)
        return h


def _siglip_can_stack(self, h) -> bool:
    if not (self.use_layer_stack and ops.stack_enabled() and h.is_cuda and h.dim() == 3 and h.is_contiguous()):
        return False
    l0 = self.layers[0]
    for ly in self.layers:
        if (ly.layer_norm1.eps != l0.layer_norm1.eps or ly.layer_norm2.eps != l0.layer_norm1.eps
                or ly.mlp.act != l0.mlp.act or ly.mlp.fc1.bias is None or ly.self_attn.q_proj.bias is None):
            return False
    return not ops.python_must_run_between(self.layers)


def _siglip_forward_stack(self, h):
    """= the loop over SiglipEncoderLayer.forward_folded, issued by aki_siglip_stack_fwd: same launches, same arguments.  The per-layer
    preparation (three folded weights out of each layer's cache, 270 pointers: ~1.1 ms of Python with the GPU idle in a one-sample prefill)
    is redone only when the cheap signature of the tower's parameters changes (ops.params_signature + the trainers' weight epoch)."""
    tb = getattr(self, "_stack_table", None)
    if tb is None:
        tb = self._stack_table = ops.LayerTable(ops.L.SiglipLayer)
    # The LIVE parameter objects, collected on every call (a cached list would keep answering for objects that load_state_dict(assign=True)
    # or `layer.mlp.fc1.weight = nn.Parameter(...)` have replaced: same address and version, other weights in use).  Sixteen dictionary
    # lookups per layer, ~0.1 ms for the tower - a tenth of the preparation the signature saves.
    live = []
    for ly in self.layers:
        at, mlp = ly.self_attn, ly.mlp
        for m in (at.q_proj, at.k_proj, at.v_proj, at.out_proj, ly.layer_norm1, ly.layer_norm2, mlp.fc1, mlp.fc2):
            pm = m._parameters
            live.append(pm["weight"])
            if pm.get("bias") is not None:
                live.append(pm["bias"])
    ep = _epoch(live)
    sig = (ep, len(self.layers), tuple(map(id, live)), ops.params_signature(live))
    if tb.sig != sig:
        rows = []
        for ly in self.layers:
            at, mlp = ly.self_attn, ly.mlp
            ps = [at.q_proj.weight, at.k_proj.weight, at.v_proj.weight, at.q_proj.bias, at.k_proj.bias, at.v_proj.bias, ly.layer_norm1.weight, ly.layer_norm1.bias]
            wqkv, bqkv, cqkv = at._prep.get("qkv_ln", ps, lambda: fold_layernorm(torch.cat([p.detach() for p in ps[:3]], 0),
                                                                                  torch.cat([p.detach() for p in ps[3:6]], 0), ly.layer_norm1), _epoch(ps))
            ln2 = ly.layer_norm2
            w1, b1, c1 = mlp._prep.get("fc1_ln", [mlp.fc1.weight, mlp.fc1.bias, ln2.weight, ln2.bias],
                                       lambda: fold_layernorm(mlp.fc1.weight.detach(), mlp.fc1.bias.detach(), ln2),
                                       _epoch([mlp.fc1.weight, mlp.fc1.bias, ln2.weight, ln2.bias]))
            w2 = mlp._prep.get("w2", [mlp.fc2.weight], lambda: ops.pad_k(mlp.fc2.weight.detach()), _epoch([mlp.fc2.weight]))
            rows.append((wqkv, bqkv, cqkv, at.out_proj.weight, at.out_proj.bias, w1, b1, c1, w2, mlp.fc2.bias))
        for r in rows:
            for t_ in r:
                if t_ is not None and not t_.is_contiguous():
                    raise ops.AkiError("siglip stack: weights must be contiguous")
        tb.get(rows)
        tb.sig = sig
    l0 = self.layers[0]
    inter = l0.mlp.fc1.weight.shape[0]
    hbuf = _fc1_buffer(h.shape[:-1], (inter + 63) // 64 * 64, h)
    return ops.siglip_stack(tb.arr, len(self.layers), h, hbuf, l0.self_attn.num_heads, inter, l0.mlp.act, l0.layer_norm1.eps, l0.self_attn.scale)


SiglipEncoder.use_layer_stack = True       # the folded inference forward as ONE C call (csrc/stack.hip) instead of 5 Python-issued launches per layer
SiglipEncoder._can_stack = _siglip_can_stack
SiglipEncoder._forward_stack = _siglip_forward_stack


class SiglipVisionTransformer(nn.Module):
    """``vision_encoder(x).last_hidden_state`` is what the reference consumes (src/vlm.py:202-203)."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.embeddings = SiglipVisionEmbeddings(config)
        self.encoder = SiglipEncoder(config)
        self.post_layernorm = nn.LayerNorm(config.hidden_size, eps=config.layer_norm_eps)

    def forward(self, pixel_values, interpolate_pos_encoding=False):
        h = self.embeddings(pixel_values, interpolate_pos_encoding=interpolate_pos_encoding)
        h = self.encoder(h)
        h = ops.layernorm(h, self.post_layernorm.weight, self.post_layernorm.bias, self.post_layernorm.eps)
        return SimpleNamespace(last_hidden_state=h, pooler_output=None)
