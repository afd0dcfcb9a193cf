"""MI355X-native Phi-3 decoder (the language model of AKI-4B).

Same module tree / parameter names as ``Phi3ForCausalLM`` (HF:phi3/modeling_phi3.py), so reference
checkpoints (``lang_model.model.layers.N.self_attn.qkv_proj.weight`` ...) load unchanged, but a decoder
layer is six HIP launches:
    rmsnorm -> [QKV projection + RoPE epilogue] -> span-driven MMA attention -> o_proj (+residual)
            -> rmsnorm -> gate_up (+SwiGLU epilogue) -> down (+residual)
The modality-mutual mask arrives as an ``ops.MaskTable`` (rectangles + valid bits), never as (B,1,L,L).
"""
from __future__ import annotations

import math
import warnings
from types import SimpleNamespace
from typing import Optional

import torch
import torch.nn.functional as F
from torch import nn

from . import ops
from . import train_ops as T
from .helpers import CausalLMOutputWithPast


def make_phi3_config(**kw):
    """A Phi3Config (HF) when transformers is importable, else a plain namespace with the same fields."""
    defaults = dict(vocab_size=32064, hidden_size=3072, intermediate_size=8192, num_hidden_layers=32,
                    num_attention_heads=32, num_key_value_heads=32, rms_norm_eps=1e-5, rope_theta=10000.0,
                    max_position_embeddings=4096, original_max_position_embeddings=4096, initializer_range=0.02,
                    pad_token_id=32000, bos_token_id=1, eos_token_id=32000, rope_scaling=None)
    defaults.update(kw)
    return SimpleNamespace(**defaults)


def _cfg_get(cfg, name, default=None):
    v = getattr(cfg, name, None)
    if v is None and hasattr(cfg, "rope_parameters") and isinstance(getattr(cfg, "rope_parameters"), dict):
        v = cfg.rope_parameters.get(name)
    return default if v is None else v


class Phi3RotaryTables(nn.Module):
    """cos/sin tables (f32) computed on the host side of the ABI, HF:phi3/modeling_phi3.py:67-124.
    Supports default RoPE and LongRoPE (short/long factors + attention scaling, as Phi-3.5 ships)."""

    def __init__(self, config):
        super().__init__()
        self.head_dim = getattr(config, "head_dim", None) or config.hidden_size // config.num_attention_heads
        self.theta = float(_cfg_get(config, "rope_theta", 10000.0))
        rs = getattr(config, "rope_scaling", None) or (getattr(config, "rope_parameters", None)
                                                       if isinstance(getattr(config, "rope_parameters", None), dict) else None)
        self.short = self.long = None
        self.orig_max = int(_cfg_get(config, "original_max_position_embeddings", config.max_position_embeddings))
        self.max_pos = int(config.max_position_embeddings)
        if rs and rs.get("rope_type", rs.get("type")) == "longrope":
            # (explicit device: the factory builds the module tree under torch.device("meta"), and these are plain attributes, not buffers)
            self.short = torch.tensor(rs["short_factor"], dtype=torch.float32, device="cpu")
            self.long = torch.tensor(rs["long_factor"], dtype=torch.float32, device="cpu")
        f = self.max_pos / self.orig_max
        self.attention_scaling = 1.0 if (self.short is None or f <= 1.0) else math.sqrt(1 + math.log(f) / math.log(self.orig_max))
        self._cache = {}

    @torch.no_grad()
    def tables(self, n_pos: int, device, seq_len: Optional[int] = None):
        """(cos, sin) f32 [n_pos, head_dim] for positions 0..n_pos-1.  LongRoPE picks its factor set from the length of
        the sequence being processed (`seq_len`, default n_pos; HF: max(position_ids)+1 > original_max -> long factors),
        which can be smaller than the table when the table is sized for a KV-cache capacity."""
        use_long = self.short is not None and (n_pos if seq_len is None else seq_len) > self.orig_max
        key = (n_pos, use_long, str(device))
        if key not in self._cache:
            d = self.head_dim
            base = self.theta ** (torch.arange(0, d, 2, dtype=torch.float32, device="cpu") / d)
            if self.short is not None:
                base = (self.long if use_long else self.short) * base
            inv = (1.0 / base).to(device)
            freqs = torch.arange(n_pos, dtype=torch.float32, device=device)[:, None] * inv[None, :]
            emb = torch.cat((freqs, freqs), dim=-1)
            if len(self._cache) > 4:
                self._cache.clear()
            self._cache[key] = ((emb.cos() * self.attention_scaling).contiguous(), (emb.sin() * self.attention_scaling).contiguous())
        return self._cache[key]


def _ag(x, *params) -> bool:
    """Take the autograd (training) path?  bf16 only: the forward reads the bf16 image of the fp32 master weights."""
    return (torch.is_grad_enabled() and x.dtype == torch.bfloat16
            and (x.requires_grad or any(p is not None and p.requires_grad for p in params)))


class AkiKVCache:
    """Per-layer K/V caches [B, H, capacity, Dh] written by the prefill (QKV+RoPE epilogue stores straight into them)
    and appended to by the decode kernels.  Lengths and positions live on the device so a decode step never syncs."""

    def __init__(self, n_layers, B, H, Dh, capacity, dtype, device):
        kv = torch.empty((2, n_layers, B, H, capacity, Dh), dtype=dtype, device=device)      # one allocation, per-layer views
        self.k, self.v = list(kv[0].unbind(0)), list(kv[1].unbind(0))
        self.capacity = capacity
        self.cache_len = torch.zeros((B,), dtype=torch.int32, device=device)   # tokens cached per sample
        self.valid_bits = None                                                  # uint64 words of the prompt's 1-D mask
        self.host_len = 0                                                       # host copy of max(cache_len)
        self.attn_ws = None                                                     # split-KV attention workspace (zeroed once)
        self.grid_keys = capacity                                               # host bound of n_keys sizing the decode grid
        self.attn_ws_rows = B
        self.chain, self.chain_sig = None, None                                 # ops.DecodeChain of the one-launch step (batch 1)
        self.chain_disabled = False                                             # set when a chained step failed its check (decode_verified)

    def get_seq_length(self, layer_idx=0):
        return int(self.cache_len.max())

    def select_rows(self, index: torch.Tensor) -> None:
        """Rows (sequences) of every per-sequence buffer gathered by `index` (int64 [B']): beam search's cache re-ordering
        (HF `_reorder_cache`) and, with repeated indices, the expansion of a prompt batch to its beams."""
        self.k = [t.index_select(0, index) for t in self.k]
        self.v = [t.index_select(0, index) for t in self.v]
        self.cache_len = self.cache_len.index_select(0, index)
        if self.valid_bits is not None:
            self.valid_bits = self.valid_bits.index_select(0, index).contiguous()
        if self.attn_ws is not None and index.numel() != self.attn_ws_rows:
            self.attn_ws = None                       # sized per sequence: rebuilt (zero-filled) by the next decode step
        self.attn_ws_rows = index.numel()

    def __getitem__(self, i):   # HF-style past_key_values[layer] -> (k, v)
        return self.k[i], self.v[i]

    def __len__(self):
        return len(self.k)


class DecodeGraph:
    """One decode step captured as a hipGraph and replayed per token: the step is ~260 short launches whose host-side
    issue cost exceeds their HBM time, so replaying removes the launch-bound gap (MI355X guide: capture launch-bound
    inner loops).  Token ids go in through a static buffer; logits come out of one.  Lengths/positions are device
    tensors advanced inside the graph, so no per-step host value is baked in - except the LongRoPE table choice, which
    is re-captured if the sequence crosses `original_max_position_embeddings`.
    Callers other than AKI.generate: a batch-1 step runs the 32 layers as one persistent launch whose dependency waits are bounded; poll
    `decode_verified(cache)` (or `cache.chain.check()`) before trusting a run of steps, as generate does - a wait that gave up leaves
    garbage logits and a sticky error word, nothing is raised by the step itself."""

    def __init__(self, lm: "Phi3ForCausalLM", cache: AkiKVCache, greedy: Optional[dict] = None):
        """greedy (optional): the arguments of ops.greedy_pick except logits / next_ids / cache_len / advance - the pick then sits INSIDE the
        replayed step and writes the next step's input ids itself: one replay per token (`step_greedy`), nothing else on the stream."""
        self.lm, self.cache = lm, cache
        B = cache.cache_len.shape[0]
        self.ids = torch.zeros((B,), dtype=torch.long, device=cache.cache_len.device)
        self.graph = None
        self.logits = None
        self._long = None
        self.greedy = greedy

    def _capture(self):
        lm, cache = self.lm, self.cache
        saved, saved_host = cache.cache_len.clone(), cache.host_len
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):               # warm-up: allocator pools and lazily-set kernel attributes
            lm.decode_step(input_ids=self.ids, past_key_values=cache)
        cur.wait_stream(side)
        cache.cache_len.copy_(saved)                # the warm-up's K/V row is overwritten by the first real step
        cache.host_len = saved_host
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            if self.greedy is None:
                self.logits = lm.decode_step(input_ids=self.ids, past_key_values=cache)
            else:                                   # the pick advances cache_len itself: one launch less per token
                self.logits = lm.decode_step(input_ids=self.ids, past_key_values=cache, advance=False)
                ops.greedy_pick(self.logits, self.ids, cache_len=cache.cache_len, advance=True, **self.greedy)
        cache.host_len = saved_host

    def step_greedy(self) -> torch.Tensor:
        """One token of a greedy generation: replay; the ids picked by the previous replay (or placed in `self.ids` by the caller) go in,
        the ids picked from this step's logits are left in `self.ids` (and in greedy['tokens'])."""
        if self.greedy is None:
            raise ops.AkiError("DecodeGraph was built without a greedy pick")
        self._ready()
        if getattr(self.cache, "chain", None) is not None:
            ops.chain_replay_on_current_stream()     # the replayed step holds a persistent chain launch: one in flight per device
        self.graph.replay()
        self.cache.host_len += 1
        return self.ids

    def _ready(self):
        if self.cache.host_len + 1 > self.cache.capacity:
            raise ops.AkiError(f"KV cache is full: {self.cache.host_len} of {self.cache.capacity} rows used; size it with "
                               "lang_model(..., use_cache=True, cache_capacity=prompt_len + max_new_tokens)")
        rot = self.lm.model.rotary_emb
        use_long = rot.short is not None and self.cache.host_len + 1 > rot.orig_max
        if self.graph is None or use_long != self._long:
            self._long = use_long
            self._capture()

    def step(self, ids: torch.Tensor) -> torch.Tensor:
        if self.greedy is not None:
            raise ops.AkiError("this DecodeGraph picks its own next ids: use step_greedy()")
        self._ready()
        self.ids.copy_(ids)
        if getattr(self.cache, "chain", None) is not None:
            ops.chain_replay_on_current_stream()
        self.graph.replay()
        self.cache.host_len += 1
        return self.logits


class Phi3RMSNorm(nn.Module):
    def __init__(self, hidden_size, eps=1e-6):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(hidden_size))
        self.variance_epsilon = eps

    def forward(self, x):
        if _ag(x, self.weight):
            return T.rmsnorm(x, self.weight, self.variance_epsilon)
        return ops.rmsnorm(x, self.weight, self.variance_epsilon)


class Phi3Attention(nn.Module):
    def __init__(self, config, layer_idx=None):
        super().__init__()
        self.layer_idx = layer_idx
        self.num_heads = config.num_attention_heads
        self.head_dim = getattr(config, "head_dim", None) or config.hidden_size // config.num_attention_heads
        if getattr(config, "num_key_value_heads", self.num_heads) != self.num_heads:
            raise NotImplementedError("AKI-4B's Phi-3.5-mini is MHA (32 kv heads); GQA is not on this path")
        self.scaling = self.head_dim ** -0.5
        op_size = 3 * self.num_heads * self.head_dim
        self.o_proj = nn.Linear(self.num_heads * self.head_dim, config.hidden_size, bias=False)
        self.qkv_proj = nn.Linear(config.hidden_size, op_size, bias=False)

    def forward(self, hidden_states, cos, sin, table, residual, position_ids=None, cache=None):
        if cache is None and _ag(hidden_states, self.qkv_proj.weight, self.o_proj.weight):
            # training: the same two kernels with their outputs kept for the backward (q, k, v, o, log-sum-exp)
            q, k, v = T.QkvRopeFn.apply(hidden_states, self.qkv_proj.weight, cos, sin, self.num_heads, position_ids)
            o = T.MmaAttnCoreFn.apply(q, k, v, table, self.scaling)
            return T.linear(o, self.o_proj.weight, None, residual)
        if cache is None:
            o = ops.mma_attn(hidden_states, self.qkv_proj.weight, cos, sin, table, self.num_heads, self.scaling, position_ids)
        else:   # prefill into the KV cache: stage 1 writes rotated K and V straight into the cache tensors
            q, k, v = ops.qkv_rope(hidden_states, self.qkv_proj.weight, cos, sin, self.num_heads, position_ids,
                                   k_out=cache.k[self.layer_idx], v_out=cache.v[self.layer_idx])
            o = ops.mma_attn_core(q, k, v, table, self.scaling)
        return ops.linear(o, self.o_proj.weight, residual=residual)

    def decode(self, h, norm, cos, sin, cache):
        """One new token per sequence: h [B, d] (residual stream, pre-norm) -> h + o_proj(attention).  Three launches:
        RMSNorm+qkv GEMV, RoPE+append+split-KV attention, o_proj GEMV with the residual add."""
        qkv = ops.decode_linear(h, self.qkv_proj.weight, norm.weight, norm.variance_epsilon)
        o = ops.decode_attn_fused(qkv, cos, sin, cache.cache_len, cache.k[self.layer_idx], cache.v[self.layer_idx], self.num_heads,
                                  self.scaling, cache.valid_bits, cache.grid_keys, cache.attn_ws)
        return ops.linear(o, self.o_proj.weight, residual=h)


class Phi3MLP(nn.Module):
    fuse_train_swiglu = True      # training forward: gate_up + SwiGLU in one launch that also keeps the pre-activations (A/B switch: tools/train_bench.py --no-fused-swiglu)

    def __init__(self, config):
        super().__init__()
        self.gate_up_proj = nn.Linear(config.hidden_size, 2 * config.intermediate_size, bias=False)
        self.down_proj = nn.Linear(config.intermediate_size, config.hidden_size, bias=False)

    def forward(self, x, residual):
        if _ag(x, self.gate_up_proj.weight, self.down_proj.weight):
            # training: the pre-activations are kept for the SwiGLU backward - the GEMM's epilogue writes them beside the activation
            a = T.gate_up_swiglu(x, self.gate_up_proj.weight) if self.fuse_train_swiglu else T.SwigluFn.apply(T.linear(x, self.gate_up_proj.weight))
            return T.linear(a, self.down_proj.weight, None, residual)
        a = ops.linear(x, self.gate_up_proj.weight, act=ops.ACT_SWIGLU)
        return ops.linear(a, self.down_proj.weight, residual=residual)

    def decode(self, h, norm):
        a = ops.decode_linear(h, self.gate_up_proj.weight, norm.weight, norm.variance_epsilon, act=ops.ACT_SWIGLU)
        return ops.linear(a, self.down_proj.weight, residual=h)


class Phi3DecoderLayer(nn.Module):
    def __init__(self, config, layer_idx):
        super().__init__()
        self.self_attn = Phi3Attention(config, layer_idx)
        self.mlp = Phi3MLP(config)
        self.input_layernorm = Phi3RMSNorm(config.hidden_size, eps=config.rms_norm_eps)
        self.post_attention_layernorm = Phi3RMSNorm(config.hidden_size, eps=config.rms_norm_eps)
        self._fp8 = None
        self._prep = ops.Prepared()

    def folds(self, h) -> bool:
        """Inference in bf16: both RMSNorms are folded into the GEMMs around them (forward_folded).  Not under parameter
        sharding (train_ops.sharded(weight)): the gain-folded weight copies are full-size and would stay resident on every
        rank, which is what FULL_SHARD exists to avoid - an evaluation forward then takes the unfolded path."""
        return (h.dtype == torch.bfloat16 and self._fp8 is None and T.CACHE_WT and not T.sharded(self.self_attn.qkv_proj.weight)
                and not _ag(h, *self.parameters()))

    def train(self, mode: bool = True):
        if mode:
            self._prep.clear()              # the gain-folded weight copies serve inference only (~5 GB for Phi-3.5-mini)
        return super().train(mode)

    def forward_folded(self, h, st, cos, sin, table, position_ids=None, cache=None):
        """The inference layer without norm launches: h is the raw residual stream and `st` its per-token 1/rms, produced by the
        epilogue of the GEMM that wrote h (o_proj / down_proj, `stats_out`).  RMSNorm(h) @ W^T = rstd * (h @ (W diag(gamma))^T), so
        the QKV and gate_up GEMMs read h itself against gain-folded weights and scale their accumulators per token.  Returns
        (h_out, stats of h_out); 4 launches per layer instead of 6."""
        at, mlp, n1, n2 = self.self_attn, self.mlp, self.input_layernorm, self.post_attention_layernorm
        wq = self._prep.get("qkv", [at.qkv_proj.weight, n1.weight], lambda: ops.fold_gain(at.qkv_proj.weight, n1.weight), T._EPOCH)
        wg = self._prep.get("gate_up", [mlp.gate_up_proj.weight, n2.weight],
                            lambda: ops.fold_gain(mlp.gate_up_proj.weight, n2.weight), T._EPOCH)
        M = h.numel() // h.shape[-1]
        if cache is None:
            o = ops.mma_attn(h, wq, cos, sin, table, at.num_heads, at.scaling, position_ids, row_scale=st.rstd)
        else:
            q, k, v = ops.qkv_rope(h, wq, cos, sin, at.num_heads, position_ids, k_out=cache.k[at.layer_idx],
                                   v_out=cache.v[at.layer_idx], row_scale=st.rstd)
            o = ops.mma_attn_core(q, k, v, table, at.scaling)
        st2 = ops.new_stats(M, h.device)
        h = ops.linear(o, at.o_proj.weight, residual=h, stats_out=st2, stats_eps=n2.variance_epsilon)
        a = ops.linear(h, wg, act=ops.ACT_SWIGLU, row_scale=st2.rstd)
        st3 = ops.new_stats(M, h.device)
        h = ops.linear(a, mlp.down_proj.weight, residual=h, stats_out=st3, stats_eps=n1.variance_epsilon)
        return h, st3

    def forward(self, h, cos, sin, table, position_ids=None, cache=None, stats=None):
        """-> h_out; with `stats` (ops.RowStats of h, see forward_folded) -> (h_out, stats of h_out)."""
        if stats is not None:
            return self.forward_folded(h, stats, cos, sin, table, position_ids, cache)
        if self._fp8 is not None and not torch.is_grad_enabled():
            return self._forward_fp8(h, cos, sin, table, position_ids, cache)
        if cache is None and _ag(h, *self.parameters()):
            # training: the norms also hand out the residual stream, so their backward sums both gradient paths in-kernel
            n1, n2 = self.input_layernorm, self.post_attention_layernorm
            x, hr = T.rmsnorm_residual(h, n1.weight, n1.variance_epsilon)
            h = self.self_attn(x, cos, sin, table, hr, position_ids, None)
            x, hr = T.rmsnorm_residual(h, n2.weight, n2.variance_epsilon)
            return self.mlp(x, hr)
        h = self.self_attn(self.input_layernorm(h), cos, sin, table, h, position_ids, cache)
        return self.mlp(self.post_attention_layernorm(h), h)

    def _forward_fp8(self, h, cos, sin, table, position_ids, cache=None):
        """BASELINE configs[4]: the four projections on the fp8 (e4m3) MFMA path with per-token activation scales and
        per-feature weight scales; RMSNorm is fused into the quantiser, RoPE / attention / residual stream stay bf16."""
        w = self._fp8
        B, L, d = h.shape
        n1, n2, at = self.input_layernorm, self.post_attention_layernorm, self.self_attn
        xq, xs = ops.quant_rows_fp8(h, n1.weight, n1.variance_epsilon)
        if cache is None:
            o = ops.mma_attn_fp8(xq, xs, *w["qkv"], cos, sin, table, B, at.num_heads, at.scaling, position_ids)
        else:   # prefill: rotated K and V go straight into the (bf16) KV cache
            q, k, v = ops.qkv_rope_fp8(xq, xs, *w["qkv"], cos, sin, B, at.num_heads, position_ids,
                                       k_out=cache.k[at.layer_idx], v_out=cache.v[at.layer_idx])
            o = ops.mma_attn_core(q, k, v, table, at.scaling)
        if w["o"] is None:      # residual_writers=False: the projections that write the residual stream stay bf16
            h = ops.linear(o.view(B, L, d), at.o_proj.weight, residual=h)
        else:
            oq, os_ = ops.quant_rows_fp8(o)
            h = ops.linear_fp8(oq, os_, *w["o"], residual=h, out_shape=(B, L, d))
        xq, xs = ops.quant_rows_fp8(h, n2.weight, n2.variance_epsilon)
        a = ops.linear_fp8(xq, xs, *w["gate_up"], act=ops.ACT_SWIGLU)
        if w["down"] is None:
            return ops.linear(a.view(B, L, -1), self.mlp.down_proj.weight, residual=h)
        aq, as_ = ops.quant_rows_fp8(a)
        return ops.linear_fp8(aq, as_, *w["down"], residual=h, out_shape=(B, L, d))

    def quantize_fp8(self, residual_writers: bool = True):
        at, mlp = self.self_attn, self.mlp
        self._fp8 = {"qkv": ops.quant_rows_fp8(at.qkv_proj.weight.detach()),
                     "o": ops.quant_rows_fp8(at.o_proj.weight.detach()) if residual_writers else None,
                     "gate_up": ops.quant_rows_fp8(mlp.gate_up_proj.weight.detach()),
                     "down": ops.quant_rows_fp8(mlp.down_proj.weight.detach()) if residual_writers else None}

    def decode(self, h, cos, sin, cache):
        if self._fp8 is not None and self._fp8["o"] is not None and h.shape[0] <= 16:
            # fp8 configuration: weight-only e4m3 GEMVs (one sequence) / skinny MFMA GEMMs (2-16) - half the bytes of the HBM-bound step, same 5 launches
            w, at = self._fp8, self.self_attn
            n1, n2 = self.input_layernorm, self.post_attention_layernorm
            qkv = ops.linear_w8(h, *w["qkv"], rms_weight=n1.weight, eps=n1.variance_epsilon)
            o = ops.decode_attn_fused(qkv, cos, sin, cache.cache_len, cache.k[at.layer_idx], cache.v[at.layer_idx], at.num_heads,
                                      at.scaling, cache.valid_bits, cache.grid_keys, cache.attn_ws)
            h = ops.linear_w8(o, *w["o"], residual=h)
            a = ops.linear_w8(h, *w["gate_up"], act=ops.ACT_SWIGLU, rms_weight=n2.weight, eps=n2.variance_epsilon)
            return ops.linear_w8(a, *w["down"], residual=h)
        h = self.self_attn.decode(h, self.input_layernorm, cos, sin, cache)
        return self.mlp.decode(h, self.post_attention_layernorm)


class Phi3Model(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = config
        self.padding_idx = getattr(config, "pad_token_id", None)
        self.embed_tokens = nn.Embedding(config.vocab_size, config.hidden_size, self.padding_idx)
        self.layers = nn.ModuleList([Phi3DecoderLayer(config, i) for i in range(config.num_hidden_layers)])
        self.norm = Phi3RMSNorm(config.hidden_size, eps=config.rms_norm_eps)
        self.rotary_emb = Phi3RotaryTables(config)
        self.skip_final_norm = False
        self.defer_final_norm = False            # set by Phi3ForCausalLM around its own call: the lm_head takes the raw stream
        self.fold_norms = True                   # bf16 inference: RMSNorms folded into the neighbouring GEMMs
        self.final_stats = None

    def forward(self, inputs_embeds, table, position_ids=None, cache=None):
        B, L, _ = inputs_embeds.shape
        seq_len = L if position_ids is None else int(position_ids.max()) + 1
        n_pos = seq_len if cache is None else max(seq_len, cache.capacity)
        cos, sin = self.rotary_emb.tables(n_pos, inputs_embeds.device, seq_len)
        if cache is not None:
            cache.host_len = seq_len
        h = inputs_embeds
        self.final_stats = None
        if self.fold_norms and all(layer.folds(h) for layer in self.layers) and not _ag(h, self.norm.weight):
            # every RMSNorm of the stack rides on the GEMM before it (statistics) and after it (gain, scale): the only pass over
            # the residual stream that is not a GEMM is this one, for the embeddings no kernel of ours produced
            if self._can_stack(h):
                h, st = self._forward_stack(h, cos, sin, table, position_ids, cache)
            else:
                st = ops.row_stats(h, self.norm.variance_epsilon)
                for layer in self.layers:
                    h, st = layer(h, cos, sin, table, position_ids, cache, stats=st)     # through __call__: module hooks (weight gathering) run
            if self.defer_final_norm:
                self.final_stats = st            # the head folds the final norm the same way
                return h
            return ops.rmsnorm(h, self.norm.weight, self.norm.variance_epsilon)
        for layer in self.layers:
            h = layer(h, cos, sin, table, position_ids, cache)
        if self.skip_final_norm:                 # the fp8 head fuses the final RMSNorm into its quantiser
            return h
        return self.norm(h)

    use_layer_stack = True                          # the folded inference forward as ONE C call (csrc/stack.hip) instead of 4-5 Python-issued launches per layer

    def train(self, mode: bool = True):
        if mode:
            self._stack_table = None                # it keeps the layers' gain-folded weight copies alive (~5 GB): inference only, like Phi3DecoderLayer._prep
        return super().train(mode)

    def _can_stack(self, h) -> bool:
        """The layer loop may run inside the library when nothing in Python has to happen between layers: no module hooks (the sharded
        trainer gathers weights there), no event tap on single launches, one epsilon for every norm, contiguous weights."""
        if not (self.use_layer_stack and ops.stack_enabled() and h.is_cuda and h.is_contiguous() and h.dim() == 3):
            return False
        eps = self.norm.variance_epsilon
        for ly in self.layers:
            if ly.input_layernorm.variance_epsilon != eps or ly.post_attention_layernorm.variance_epsilon != eps:
                return False
        return not ops.python_must_run_between(self.layers)        # hooks on the layers or their sub-modules, instance-level forward overrides

    def _forward_stack(self, h, cos, sin, table, position_ids, cache):
        """= the loop over Phi3DecoderLayer.forward_folded, issued by aki_decoder_stack_fwd: same launches, same arguments.
        The per-layer preparation (gain-folded weights out of each layer's cache, 192 pointers) costs ~0.8 ms of Python - during which the
        GPU of a one-sample prefill sits idle - so it is redone only when a cheap signature of the weights changes: (address, in-place
        version) of all 6 x n_layers tensors, the trainers' weight epoch, the module-level version bumped by _apply / load_state_dict."""
        tb = getattr(self, "_stack_table", None)
        if tb is None:
            tb = self._stack_table = ops.LayerTable(ops.L.DecoderLayer)
            tb.params = None
        if tb.params is None or tb.params[0] != self._weights_version:
            flat = []
            for ly in self.layers:
                flat += [ly.self_attn.qkv_proj.weight, ly.input_layernorm.weight, ly.self_attn.o_proj.weight, ly.mlp.gate_up_proj.weight,
                         ly.post_attention_layernorm.weight, ly.mlp.down_proj.weight]
            tb.params = (self._weights_version, flat)
            tb.sig = None
        sig = (T._EPOCH, self._weights_version, len(self.layers), ops.params_signature(tb.params[1]))
        if tb.sig != sig:
            rows = []
            for ly in self.layers:
                at, mlp, n1, n2 = ly.self_attn, ly.mlp, ly.input_layernorm, ly.post_attention_layernorm
                wq = ly._prep.get("qkv", [at.qkv_proj.weight, n1.weight], lambda: ops.fold_gain(at.qkv_proj.weight, n1.weight), T._EPOCH)
                wg = ly._prep.get("gate_up", [mlp.gate_up_proj.weight, n2.weight], lambda: ops.fold_gain(mlp.gate_up_proj.weight, n2.weight), T._EPOCH)
                rows.append((wq, at.o_proj.weight, wg, mlp.down_proj.weight, None, None))
            for r in rows:
                for t_ in r[:4]:
                    if not t_.is_contiguous():
                        raise ops.AkiError("decoder stack: weights must be contiguous")
            tb.get(rows)
            tb.sig = sig
        arr = tb.arr
        for i, ly in enumerate(self.layers):                # the KV cache is this call's: 64 pointer stores
            arr[i].k_cache = None if cache is None else cache.k[i].data_ptr()
            arr[i].v_cache = None if cache is None else cache.v[i].data_ptr()
        at0 = self.layers[0].self_attn
        return ops.decoder_stack(arr, len(self.layers), h, cos, sin, table, at0.num_heads, at0.head_dim,
                                 self.layers[0].mlp.down_proj.weight.shape[1], at0.scaling, self.norm.variance_epsilon, position_ids,
                                 0 if cache is None else cache.capacity)

    def decode(self, inputs_embeds, cache, advance: bool = True):
        """inputs_embeds [B, d]: the embeddings of the tokens appended at index cache.cache_len[b].  advance=False leaves cache_len to the
        caller (ops.greedy_pick folds the increment into its launch)."""
        if cache.host_len + 1 > cache.capacity and not torch.cuda.is_current_stream_capturing():
            # host_len is a host-side upper bound of max(cache_len): the guard costs no sync.  The append kernels write row
            # cache_len[b] and read cos/sin row cache_len[b] unconditionally - one step further corrupts the next (b, h) slab.
            raise ops.AkiError(f"KV cache is full: {cache.host_len} of {cache.capacity} rows used; size it with "
                               "lang_model(..., use_cache=True, cache_capacity=prompt_len + max_new_tokens)")
        cache.host_len += 1                         # host-side upper bound of max(cache_len)+1: no device sync per step
        cos, sin = self.rotary_emb.tables(cache.capacity, inputs_embeds.device, cache.host_len)
        # position of the new token = its cache row = number of tokens before it (cache.cache_len, on the device)
        if cache.attn_ws is None:
            cache.attn_ws = ops.decode_attn_workspace(inputs_embeds.shape[0], cache.k[0].shape[1], cache.k[0].shape[3],
                                                      cache.capacity, inputs_embeds.device)
        cache.grid_keys = cache.capacity if torch.cuda.is_current_stream_capturing() else min(cache.capacity, cache.host_len)
        h = inputs_embeds
        chain = self._decode_chain(h, cache)
        if chain is not None:                       # 1..8 sequences: the 32 layers as ONE launch (decode_chain.hip)
            h = chain.step(h.reshape(chain.batch, -1).contiguous(), cos, sin, cache.cache_len, cache.valid_bits, cache.grid_keys)
        else:
            for layer in self.layers:
                h = layer.decode(h, cos, sin, cache)
        if advance:
            cache.cache_len += 1
        return h                                    # PRE-norm: the head applies self.norm inside its GEMV

    _weights_version = 0                            # bumped whenever parameters may have been re-allocated (part of the chain's signature)

    def _apply(self, fn, *a, **kw):
        self._weights_version += 1
        return super()._apply(fn, *a, **kw)

    def load_state_dict(self, *a, **kw):
        self._weights_version += 1
        return super().load_state_dict(*a, **kw)

    def _load_from_state_dict(self, *a, **kw):      # reached when a PARENT module loads (assign=True swaps the parameter objects)
        self._weights_version += 1
        return super()._load_from_state_dict(*a, **kw)

    use_decode_chain = True                         # False: the five-launch-per-layer path (A/B and the bit-identity tests)
    use_decode_chain_batched = False                # True (LAB library only: the product library does not carry the kernel and answers AKI_ERR_UNSUPPORTED):
                                                    # 2..8 sequences on the batched chain - bit-identical to the five launches per layer, measured NOT faster
                                                    # (3.29-3.46 vs 3.27-3.34 ms per step at batch 8, EXPERIMENTS.md round 5)
    decode_chain_w8 = True                          # e4m3 weights: one batch per workgroup, 1.37 ms per token against 1.46 on five launches
                                                    # (1.55 vs 1.46 ms per token: half the bytes, the same dependency latencies)

    def _decode_chain(self, h, cache):
        """The one-launch decode step when it applies: one sequence, bf16 stream, Phi-3.5-mini's dimensions, every layer either
        bf16 or fully e4m3-quantised (the fp8 configuration's weight-only GEMVs).  Built once per (weights, KV cache)."""
        B = h.shape[0]
        if (not self.use_decode_chain or cache.chain_disabled or B > (8 if self.use_decode_chain_batched else 1) or h.dtype != torch.bfloat16
                or not h.is_cuda):
            return None
        l0, ll = self.layers[0], self.layers[-1]
        chain = getattr(cache, "chain", None)
        # a cheap signature per step (module attribute lookups cost ~1 us each; the full scan below runs only when it changes):
        # re-allocated weights (model.to / a new quantisation) or KV tensors (beam re-ordering) move these pointers
        # (ends of the stack), and everything that re-allocates parameters in between - Module._apply (.to / .half / .cuda), load_state_dict,
        # a new quantisation - bumps `_weights_version`.  Assigning a new tensor to one middle layer's `.data` by hand is not seen:
        # build a fresh cache (a new prefill) after surgery of that kind.
        sig = (l0.self_attn.qkv_proj.weight.data_ptr(), ll.mlp.down_proj.weight.data_ptr(), self._weights_version, cache.k[0].data_ptr(),
               cache.k[-1].data_ptr(), len(self.layers), B)
        if chain is not None and chain.sig == sig:
            return chain
        if chain is None and getattr(cache, "chain_sig", None) == sig:
            return None                             # this (model, cache) pair was found unsupported before
        cache.chain_sig, cache.chain = sig, None
        at, mlp = l0.self_attn, l0.mlp
        sup = ops.DecodeChain.SUPPORTED
        if (at.head_dim != sup["Dh"] or h.shape[-1] != sup["d"] or mlp.down_proj.weight.shape[1] != sup["F"]
                or any(n.variance_epsilon != l0.input_layernorm.variance_epsilon for ly in self.layers
                       for n in (ly.input_layernorm, ly.post_attention_layernorm))):
            return None
        w8 = l0._fp8 is not None and l0._fp8["o"] is not None
        if any((ly._fp8 is not None and ly._fp8["o"] is not None) != w8 for ly in self.layers):
            return None
        if l0._fp8 is not None and not w8:
            return None                             # qkv / gate_up only in e4m3: the mixed per-layer path
        if w8 and not self.decode_chain_w8:
            return None
        if B > 1 and (w8 or at.num_heads != 32 or cache.k[0].shape[0] != B or not cache.k[0].is_contiguous()):
            return None                             # the batched chain: bf16 weights, caches [B, H, capacity, 96]
        rows = []
        for ly in self.layers:
            a_, m_ = ly.self_attn, ly.mlp
            if w8:
                f = ly._fp8
                rows.append((f["qkv"][0], f["o"][0], f["gate_up"][0], f["down"][0], ly.input_layernorm.weight, ly.post_attention_layernorm.weight,
                             f["qkv"][1], f["o"][1], f["gate_up"][1], f["down"][1]))
            else:
                rows.append((a_.qkv_proj.weight, a_.o_proj.weight, m_.gate_up_proj.weight, m_.down_proj.weight, ly.input_layernorm.weight,
                             ly.post_attention_layernorm.weight, None, None, None, None))
        chain = cache.chain = ops.DecodeChain(rows, list(cache.k), list(cache.v), at.num_heads, at.head_dim, h.shape[-1],
                                              mlp.down_proj.weight.shape[1], cache.capacity, at.scaling, l0.input_layernorm.variance_epsilon,
                                              h.device, w8, batch=B)
        chain.sig = sig
        return chain


class Phi3ForCausalLM(nn.Module):
    """Drop-in for the HF class on the AKI path: ``lang_model(inputs_embeds=..., attention_mask=<MaskTable>, labels=...)``."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.model = Phi3Model(config)
        self.vocab_size = config.vocab_size
        self.lm_head = nn.Linear(config.hidden_size, config.vocab_size, bias=False)
        self.generation_config = None     # set from the checkpoint's generation_config.json by the loaders (eos ids for generate)
        self._prep = ops.Prepared()

    def train(self, mode: bool = True):
        if mode:
            self._prep.clear()              # the gain-folded head copy serves inference only
        return super().train(mode)

    # --- the accessors src/vlm.py:48,80-99 relies on -----------------------------------------------------
    def get_input_embeddings(self):
        return self.model.embed_tokens

    def set_input_embeddings(self, value):
        self.model.embed_tokens = value

    def get_output_embeddings(self):
        return self.lm_head

    def set_output_embeddings(self, new_embeddings):
        self.lm_head = new_embeddings

    def enable_fp8(self, enable: bool = True, head: bool = True, residual_writers: bool = True):
        """Quantise the decoder's projection weights (and the lm_head) to e4m3 once; inference forwards without a KV cache
        then run their GEMMs on the fp8 MFMA path.  The bf16 weights stay in place (decode, training, disable).
        Every e4m3 GEMM output carries a few percent of rounding noise (3 mantissa bits on both operands); it accumulates in
        the residual stream over 32 layers and lands directly on the logits through the head (measured at full depth:
        tests/test_full_depth_gpu.py).  `head=False` keeps the lm_head in bf16, `residual_writers=False` also
        o_proj and down_proj (the projections whose output IS the residual stream): qkv and gate_up - 62 % of the decoder's
        GEMM work - then still run on the fp8 MFMA path."""
        self.model._weights_version += 1
        if not enable:
            for layer in self.model.layers:
                layer._fp8 = None
            self._fp8_head = None
            return self
        if self.model.embed_tokens.weight.dtype != torch.bfloat16:
            raise ops.AkiError("enable_fp8: the model must hold bf16 weights")
        for layer in self.model.layers:
            layer.quantize_fp8(residual_writers)
        self._fp8_head = None
        if head:
            hd = self.lm_head
            if type(hd) is nn.Linear:
                w, b, n = hd.weight.detach(), hd.bias, hd.weight.shape[0]
            else:
                w, b, n = hd._fused_weight()
            self._fp8_head = (*ops.quant_rows_fp8(w), b, n)
        return self

    def _head(self, h):
        if getattr(self, "_fp8_head", None) is not None and not torch.is_grad_enabled() and getattr(self, "_pre_norm_h", None) is not None:
            wq, ws, b, n = self._fp8_head
            hq, hs = ops.quant_rows_fp8(self._pre_norm_h, self.model.norm.weight, self.model.norm.variance_epsilon)
            self._pre_norm_h = None
            return ops.linear_fp8(hq, hs, wq, ws, bias=b, out_shape=(*h.shape[:-1], wq.shape[0]))[..., :n]
        if type(self.lm_head) is nn.Linear:
            return ops.linear(h, self.lm_head.weight, bias=self.lm_head.bias)
        return self.lm_head(h)   # DecoupledLinear (src/vlm.py:88-99): one HIP GEMM / GEMV over the fused weight

    def _head_folded(self, h, st):
        """lm_head(norm(h)) with the final RMSNorm folded: gain into a cached copy of the head weight, 1/rms into the epilogue."""
        norm, head = self.model.norm, self.lm_head
        if type(head) is nn.Linear:
            w = self._prep.get("head", [head.weight, norm.weight], lambda: ops.fold_gain(head.weight, norm.weight), T._EPOCH)
            return ops.linear(h, w, bias=head.bias, row_scale=st.rstd)
        w0, b, n = head._fused_weight()
        w = self._prep.get("head", [w0, norm.weight], lambda: ops.fold_gain(w0, norm.weight), (T._EPOCH, head._fused[0]))
        return ops.linear(h, w, bias=b, row_scale=st.rstd)[..., :n]

    def forward(self, input_ids=None, attention_mask=None, inputs_embeds=None, labels=None, position_ids=None,
                use_cache=False, past_key_values=None, cache_capacity=None, last_token_logits=False, **kwargs):
        """Prefill / full forward.  With use_cache=True the returned past_key_values is an AkiKVCache holding the
        rotated keys and the values of every layer (capacity = cache_capacity or L + 256).
        last_token_logits=True (prefill of `generate`, which reads nothing else; HF's `logits_to_keep=1` for a right-padded batch): logits is
        [B, 1, V'] - the head runs on each sample's last valid token only, a weight-streaming GEMV instead of an L-row GEMM against the
        197 MB head."""
        if past_key_values is not None:
            return self._continue(input_ids, inputs_embeds, past_key_values, labels)
        if last_token_logits and labels is not None:
            raise ValueError("last_token_logits=True keeps one row of logits per sample; a loss over `labels` needs all of them")
        if last_token_logits and not use_cache:
            raise ValueError("last_token_logits=True is the prefill of a generation: pass use_cache=True (without a cache the flag used to be ignored silently)")
        if inputs_embeds is None:
            inputs_embeds = self.model.embed_tokens(input_ids)
        B, L, _ = inputs_embeds.shape
        table = attention_mask
        if table is None:
            table = ops.MaskTable.causal(B, L, inputs_embeds.device)
        elif isinstance(table, torch.Tensor):
            table = mask_table_from_tensor(table, L)
        cache = None
        if use_cache:
            cfg = self.config
            H = cfg.num_attention_heads
            Dh = getattr(cfg, "head_dim", None) or cfg.hidden_size // H
            cache = AkiKVCache(len(self.model.layers), B, H, Dh, int(cache_capacity or (L + 256)), inputs_embeds.dtype,
                               inputs_embeds.device)
        fp8_head = getattr(self, "_fp8_head", None) is not None and not torch.is_grad_enabled()
        self.model.skip_final_norm = fp8_head
        self.model.defer_final_norm = True
        h = self.model(inputs_embeds, table, position_ids, cache)
        self.model.skip_final_norm = self.model.defer_final_norm = False
        head_stats, self.model.final_stats = self.model.final_stats, None     # not None: h is the raw stream, norm folded into the head
        self._pre_norm_h = h if fp8_head else None
        if head_stats is not None and _ag(h, *self.lm_head.parameters()):
            # frozen decoder under a trainable head: the head's autograd path wants the normalised stream itself
            h, head_stats = ops.rmsnorm(h, self.model.norm.weight, self.model.norm.variance_epsilon), None
        if labels is not None and cache is None and _ag(h, *self.lm_head.parameters()):
            # training: lm_head + shifted cross-entropy chunk by chunk - no [B, L, V] logits tensor, no concatenated head
            # weight (so the output carries no logits; train/losses.py:110-115 only reads [0] = loss)
            loss = T.fused_head_ce(h, self.lm_head, labels, chunk=getattr(self, "head_chunk_rows", 2688))
            return CausalLMOutputWithPast(loss=loss, logits=None, past_key_values=None)
        counts = None if cache is None else table.token_counts(B, h.device)
        if last_token_logits and cache is not None and labels is None and h.dtype == torch.bfloat16 and (head_stats is not None or fp8_head):
            # h is the raw stream here (final norm deferred to the head): gather the B rows, then the decode step's head
            rows = (counts.long() - 1).clamp_(min=0) + torch.arange(B, device=h.device) * L
            logits = self._head_rows(h.reshape(B * L, -1).index_select(0, rows)).unsqueeze(1)
            self._pre_norm_h = None
        else:
            logits = self._head(h) if head_stats is None else self._head_folded(h, head_stats)
            if last_token_logits and cache is not None:
                logits = logits[torch.arange(B, device=h.device), (counts.long() - 1).clamp_(min=0)].unsqueeze(1)
        if cache is not None:
            cache.cache_len.copy_(counts)
            if table.col_valid_bits is not None:
                # decode appends right after each sample's last valid token (overwriting stacking padding), so only holes INSIDE the prompt stay masked
                nw = table.col_valid_bits.shape[1]
                idx = torch.arange(nw * 64, device=h.device, dtype=torch.int32)[None, :]
                beyond = (idx >= cache.cache_len[:, None]).view(B, nw, 64)
                weights = (torch.ones(64, dtype=torch.int64, device=h.device) << torch.arange(64, device=h.device))
                beyond_bits = (beyond.to(torch.int64) * weights).sum(-1)      # wraps into the sign bit exactly like uint64
                cache.valid_bits = (table.col_valid_bits | beyond_bits).contiguous()
        loss = None
        if labels is not None:
            loss = causal_lm_loss(logits, labels)
        return CausalLMOutputWithPast(loss=loss, logits=logits, past_key_values=cache)

    def _continue(self, input_ids, inputs_embeds, cache, labels=None):
        """Text-only continuation from an AkiKVCache - the `past_key_values is not None` call of the reference
        (src/vlm.py:463-475 -> src/aki.py:125-130 with `input_ids` only): the new tokens attend to everything cached plus
        their own causal prefix, i.e. T teacher-forced decode steps.  Returns logits [B, T, V'] and the same cache."""
        if not isinstance(cache, AkiKVCache):
            raise ops.AkiError("past_key_values must be the AkiKVCache returned by a use_cache=True forward of this model")
        if labels is not None:
            raise NotImplementedError("labels with past_key_values: the loss is defined on the full forward only")
        if inputs_embeds is None:
            inputs_embeds = self.get_input_embeddings()(input_ids)
        if inputs_embeds.dim() == 2:
            inputs_embeds = inputs_embeds[:, None]
        T_new = inputs_embeds.shape[1]
        steps = [self.decode_step(inputs_embeds=inputs_embeds[:, t], past_key_values=cache) for t in range(T_new)]
        if not self.decode_verified(cache):          # a chained step failed its check: the same steps again, five launches per layer
            cache.cache_len -= T_new
            cache.host_len -= T_new
            steps = [self.decode_step(inputs_embeds=inputs_embeds[:, t], past_key_values=cache) for t in range(T_new)]
        return CausalLMOutputWithPast(loss=None, logits=torch.stack(steps, dim=1), past_key_values=cache)

    def decode_verified(self, cache) -> bool:
        """True when every decode step taken on `cache` so far is valid.  The one-launch decode chain (one sequence; decode_chain.hip) bounds
        its dependency waits, and a wait that gives up leaves garbage in that step's output and a sticky error word: this reads the word (a
        device synchronisation - callers do it where they synchronise anyway).  On an error the chain is switched off for this cache (later
        steps take the five-launch-per-layer path), a warning is issued and False is returned: the caller must discard every step since its
        last verified point and run them again (rewind cache.cache_len / cache.host_len; the K/V rows are simply overwritten)."""
        chain = getattr(cache, "chain", None)
        if chain is None:
            return True
        code = chain.error_code()
        if code == 0:
            return True
        cache.chain, cache.chain_disabled = None, True
        warnings.warn(f"decode chain: a dependency wait gave up (layer {code >> 8}, phase {code & 255}); the steps since the last verified "
                      "token are decoded again on the per-layer path, which this KV cache now stays on", RuntimeWarning, stacklevel=2)
        return False

    def decode_step(self, input_ids=None, inputs_embeds=None, past_key_values=None, advance: bool = True):
        """One greedy-decoding step: new token ids [B] (or their embeddings [B, d]) -> logits [B, V'].  After the prefill
        the reference's mask is all ones (src/aki_generation.py:58-62): the token attends to everything cached."""
        if inputs_embeds is None:
            inputs_embeds = self.get_input_embeddings()(input_ids)
        h = self.model.decode(inputs_embeds.reshape(inputs_embeds.shape[0], -1), past_key_values, advance=advance)
        return self._head_rows(h)

    def _head_rows(self, h):
        """lm_head(norm(h)) for a few rows of the RAW residual stream [B, d] (a decode step; the last prompt token of a prefill): the final
        RMSNorm is applied inside the weight-streaming GEMV."""
        norm = self.model.norm
        if getattr(self, "_fp8_head", None) is not None and h.shape[0] <= 16:
            wq, ws, b, n = self._fp8_head
            return ops.linear_w8(h, wq, ws, bias=b, rms_weight=norm.weight, eps=norm.variance_epsilon)[..., :n]
        if type(self.lm_head) is nn.Linear:
            return ops.decode_linear(h, self.lm_head.weight, norm.weight, norm.variance_epsilon, bias=self.lm_head.bias)
        return self.lm_head.forward_normed(h, norm.weight, norm.variance_epsilon)


def causal_lm_loss(logits, labels, ignore_index=-100):
    """HF ForCausalLMLoss: shift by one, mean cross entropy over non-ignored targets, computed in f32.  bf16 logits on the
    GPU go through the HIP cross-entropy kernel (forward only, read in place through their row stride); the exact-f32 parity
    path keeps torch's."""
    if (logits.is_cuda and logits.dtype == torch.bfloat16 and ignore_index == -100 and logits.dim() == 3 and logits.stride(2) == 1
            and logits.stride(0) == logits.shape[1] * logits.stride(1) and logits.stride(1) % 2 == 0):
        B, L, n = logits.shape
        ld = logits.stride(1)
        rows = torch.as_strided(logits, (B, L, ld), (L * ld, ld, 1)) if ld != n else logits
        return T.ce_loss(rows, labels.to(logits.device), n, want_grad=False)[0]
    lg = logits[:, :-1].float()
    tg = labels[:, 1:].to(lg.device)
    return F.cross_entropy(lg.reshape(-1, lg.shape[-1]), tg.reshape(-1), ignore_index=ignore_index)


def mask_table_from_tensor(mask: torch.Tensor, L: int) -> ops.MaskTable:
    """Accept what a reference caller passes as `attention_mask`:
      * (B, L) 0/1 padding mask            -> causal + valid bits (HF semantics);
      * (B, 1, L, L) 0/1 dense mask        -> the hand-off type of the reference (`_prepare_inputs_for_forward`,
        src/vlm.py:589-603, consumed at src/aki.py:125-130).  It is converted ON THE DEVICE into the rectangle table and the
        conversion is verified bit for bit (ops.mask_to_table); masks outside the modality-mutual family raise AkiError.
    The attention kernels never see an L x L tensor either way."""
    if mask.dim() == 2:
        return ops.MaskTable.from_host([[(0, 0, 0, 0)]] * mask.shape[0], mask.detach().cpu().numpy(), None, mask.device)
    if mask.dim() == 4:
        if mask.shape[-1] != L or mask.shape[-2] != L:
            raise ValueError(f"attention_mask {tuple(mask.shape)} does not match the sequence length {L}")
        if mask.is_floating_point() and bool((mask < 0).any()):
            raise ValueError("additive (already inverted) 4-D masks are not accepted: pass the 0/1 mask the reference builds "
                             "(src/vlm.py:438-443) - the finfo.min inversion of transformers 4.41.2 is part of the kernel")
        return ops.mask_to_table(mask.detach())
    raise ValueError(f"attention_mask must be an ops.MaskTable, a (B,L) or a (B,1,L,L) tensor; got {tuple(mask.shape)}")
