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


class _Prepared:
    """Cache of one-time weight transforms (K padding, QKV concatenation), invalidated when a parameter changes."""

    def __init__(self):
        self._c = {}

    def get(self, name, params, fn):
        key = tuple((p.data_ptr(), p._version, p.dtype, p.device) for p in params)
        hit = self._c.get(name)
        if hit is None or hit[0] != key:
            with torch.no_grad():
                self._c[name] = (key, fn())
        return self._c[name][1]


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

    def forward(self, x, residual):
        N, L, E = x.shape
        ps = [self.q_proj.weight, self.k_proj.weight, self.v_proj.weight, self.q_proj.bias, self.k_proj.bias, self.v_proj.bias]
        wqkv, bqkv = self._prep.get("qkv", ps, lambda: (torch.cat([p.detach() for p in ps[:3]], 0).contiguous(),
                                                        torch.cat([p.detach() for p in ps[3:]], 0).contiguous()))
        qkv = ops.linear(x, wqkv, bias=bqkv).view(N, L, 3, self.num_heads, self.head_dim)
        a = ops.attention(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], self.scale)   # strided views, no copies
        return ops.linear(a, self.out_proj.weight, bias=self.out_proj.bias, residual=residual)


_HBUF = {}      # (leading shape, padded width, dtype, device, stream) -> K-padded fc1 output buffer shared by all layers


class SiglipMLP(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.fc1 = nn.Linear(config.hidden_size, config.intermediate_size)
        self.fc2 = nn.Linear(config.intermediate_size, config.hidden_size)
        self.act = ops.ACT_GELU_TANH if "tanh" in config.hidden_act else ops.ACT_GELU_ERF
        self._prep = _Prepared()

    def forward(self, x, residual):
        inter = self.fc1.weight.shape[0]
        Kp = (inter + 63) // 64 * 64
        w2 = self._prep.get("w2", [self.fc2.weight], lambda: ops.pad_k(self.fc2.weight.detach()))
        lead = x.shape[:-1]
        # fc1 writes into a K-padded buffer; the pad columns must be finite zeros for fc2 (zero weights there).  The buffer
        # is kept per shape: fc1 overwrites columns [:inter] every call and nothing ever writes the pad columns, so they are
        # zeroed once instead of a 40 MB fill per layer and step (stream-ordered reuse: the next fc1 runs after this fc2).
        key = (tuple(lead), Kp, x.dtype, x.device, torch.cuda.current_stream().cuda_stream)
        hbuf = _HBUF.get(key)
        if hbuf is None:
            if len(_HBUF) > 8:
                _HBUF.clear()
            hbuf = _HBUF[key] = torch.zeros((*lead, Kp), dtype=x.dtype, device=x.device)
        ops.linear(x, self.fc1.weight, bias=self.fc1.bias, act=self.act, out=hbuf[..., :inter])
        return ops.linear(hbuf, w2, bias=self.fc2.bias, residual=residual)


class SiglipEncoderLayer(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.layer_norm1 = nn.LayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.self_attn = SiglipAttention(config)
        self.layer_norm2 = nn.LayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.mlp = SiglipMLP(config)

    def forward(self, h):
        h = self.self_attn(ops.layernorm(h, self.layer_norm1.weight, self.layer_norm1.bias, self.layer_norm1.eps), h)
        return self.mlp(ops.layernorm(h, self.layer_norm2.weight, self.layer_norm2.bias, self.layer_norm2.eps), h)


class SiglipEncoder(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.layers = nn.ModuleList([SiglipEncoderLayer(config) for _ in range(config.num_hidden_layers)])

    def forward(self, h):
        for layer in self.layers:
            h = layer(h)
        return h


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
