"""Torch-tensor front end of the C ABI (include/aki_mi355x.h).

PyTorch is plumbing here: device memory, the current HIP stream and autograd bookkeeping.  Every
function below hands raw device pointers to libaki_mi355x.so; there is NO eager/CPU fallback -
CPU tensors or a missing library raise :class:`aki_amd._lib.AkiError`.
"""
from __future__ import annotations

import ctypes as C
import threading
from dataclasses import dataclass
from typing import Optional, Tuple

import torch

from . import _lib as L
from ._lib import AkiError

ACT_NONE, ACT_GELU_ERF, ACT_GELU_TANH, ACT_SWIGLU = L.AKI_ACT_NONE, L.AKI_ACT_GELU_ERF, L.AKI_ACT_GELU_TANH, L.AKI_ACT_SWIGLU
DEAD_ROWS_ZERO, DEAD_ROWS_UNIFORM = L.AKI_DEAD_ROWS_ZERO, L.AKI_DEAD_ROWS_UNIFORM


def _dt(t: torch.Tensor) -> int:
    if t.dtype == torch.bfloat16:
        return L.AKI_DT_BF16
    if t.dtype == torch.float32:
        return L.AKI_DT_F32
    raise AkiError(f"unsupported dtype {t.dtype}: the AKI HIP path computes in bf16 or f32")


def _dev(*ts: Optional[torch.Tensor]) -> torch.device:
    dev = None
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise AkiError("the AKI MMA path runs on MI355X only: got a CPU tensor and there is no CPU fallback")
        dev = dev or t.device
        if t.device != dev:
            raise AkiError("tensors on different devices")
    return dev


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _rows2d(t: torch.Tensor) -> torch.Tensor:
    """View as [rows, cols] with unit inner stride (copy only if the layout forces it)."""
    t2 = t.reshape(-1, t.shape[-1])
    if t2.stride(-1) != 1 or (t2.shape[0] > 1 and t2.stride(0) < t2.shape[1]):
        t2 = t2.contiguous()
    return t2


def _ws(nbytes: int, dev) -> torch.Tensor:
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=dev)


class EventTap:
    """Optional per-kernel timing with HIP events on the stream the kernels are launched on (torch's current
    stream).  bench.py installs one for the timed region; when no tap is installed the cost is one `is None` test."""

    def __init__(self, tags=None, select=None, every=1):
        self.tags = tags          # None = every tagged launch
        self.select = select      # optional predicate on the full tag (e.g. only one GEMM shape): every bracketed launch
        self.events = {}          # costs two event records on the stream (~6 us of dispatch gap each on MI355X, see
        self.every = every        # tools/trace_gaps.py), so time only what is reported - and only every `every`-th launch of it
        self.calls = {}

    def want(self, tag) -> bool:
        if self.tags is not None and tag[0] not in self.tags:
            return False
        return self.select is None or bool(self.select(tag))

    def begin(self, tag):
        n = self.calls[tag] = self.calls.get(tag, 0) + 1
        if (n - 1) % self.every:
            return None
        a = torch.cuda.Event(enable_timing=True)
        b = torch.cuda.Event(enable_timing=True)
        self.events.setdefault(tag, []).append((a, b))
        a.record()
        return b

    def summary(self):
        torch.cuda.synchronize()
        return {tag: (self.calls[tag], sum(a.elapsed_time(b) for a, b in ev) / len(ev)) for tag, ev in self.events.items()}


class Prepared:
    """Cache of one-time weight transforms (K padding, QKV concatenation, folded norm gains), rebuilt when a parameter changes
    (version counter, storage, dtype, device) or when the trainer announces raw-pointer weight writes (`epoch`)."""

    def __init__(self):
        self._c = {}

    def get(self, name, params, fn, epoch=0):
        key = (epoch,) + tuple((p.data_ptr(), p._version, p.dtype, p.device) for p in params)
        hit = self._c.get(name)
        if hit is None or hit[0] != key:
            with torch.no_grad():
                self._c[name] = (key, fn())
        return self._c[name][1]

    def clear(self) -> None:
        """Drop every cached transform (model.train(): the folded copies are only read by inference forwards)."""
        self._c.clear()


def fold_gain(w: torch.Tensor, gain: torch.Tensor) -> torch.Tensor:
    """W' = W * diag(gain), rounded once to W's dtype: x_normed(gain) @ W^T == (x * rstd) @ W'^T up to that rounding."""
    ct = torch.float64 if w.dtype == torch.float64 else torch.float32          # computed in f32 (f64 stays f64), rounded once
    return (w.detach().to(ct) * gain.detach().to(ct)[None, :]).to(w.dtype).contiguous()


@dataclass
class RowStats:
    """Per-row statistics of an activation [M, N], produced by the epilogue of the GEMM that wrote it (or by `row_stats`) and
    consumed by the next GEMM as `row_scale` / `row_shift`: the block's pre-norm then costs no launch of its own."""
    rstd: torch.Tensor                      # f32 [M]: 1/sqrt(mean(y^2) + eps) (RMSNorm) or 1/sqrt(var(y) + eps) (LayerNorm)
    mean: Optional[torch.Tensor] = None     # f32 [M] (LayerNorm only)


_STATS_WS = {}


def _stats_ws(M: int, n_out: int, dev) -> torch.Tensor:
    """Workspace of the producer side (arrival counters + partial sums): zero-filled once per (device, stream).  The counter
    area at its front has one size for every M (include/aki_mi355x.h), so a buffer serves launches of any size in any order."""
    need = int(L.load().aki_linear_stats_workspace_bytes(M, n_out))
    key = (torch.device(dev).index, torch.cuda.current_stream().cuda_stream)
    hit = _STATS_WS.get(key)
    if hit is None or hit.numel() < need:
        hit = _STATS_WS[key] = torch.zeros(max(need, 8 << 20), dtype=torch.uint8, device=dev)
    return hit


_SPLITK_WS = {}
SPLITK_WS_MIN_BYTES = 0          # tools / tests: a floor for the split-K workspace (the lab library's forced splits need more than the planner's)
SPLITK_MAX_M = 2048              # rows above which linear() hands no split-K workspace over (the planner splits below 1024 only; tools raise it)


def _splitk_ws(M: int, N: int, K: int, dev) -> Optional[torch.Tensor]:
    """Split-K workspace (tickets + f32 partial tiles; include/aki_mi355x.h): zero-filled once per (device, stream), None when the
    library never splits this shape."""
    need = max(int(L.load().aki_linear_splitk_workspace_bytes(M, N, K)), SPLITK_WS_MIN_BYTES)
    if need == 0:
        return None
    key = (torch.device(dev).index, torch.cuda.current_stream().cuda_stream)
    hit = _SPLITK_WS.get(key)
    if hit is None or hit.numel() < need:
        hit = _SPLITK_WS[key] = torch.zeros(max(need, 32 << 20), dtype=torch.uint8, device=dev)
    return hit


def new_stats(M: int, dev, ln: bool = False) -> RowStats:
    return RowStats(torch.empty((M,), dtype=torch.float32, device=dev), torch.empty((M,), dtype=torch.float32, device=dev) if ln else None)


def row_stats(x: torch.Tensor, eps: float, ln: bool = False) -> RowStats:
    """Statistics of a tensor no GEMM of this library produced (the first block's input): one HBM pass, no output tensor."""
    dev = _dev(x)
    x2 = _rows2d(x)
    st = new_stats(x2.shape[0], dev, ln)
    L.check(L.load().aki_row_stats(_ptr(x2), x2.shape[0], x2.shape[1], x2.stride(0), float(eps), _ptr(st.rstd), _ptr(st.mean), _dt(x),
                                   _stream()), "aki_row_stats")
    return st


_TAP: Optional[EventTap] = None


def set_event_tap(tap: Optional[EventTap]) -> None:
    global _TAP
    _TAP = tap


# ------------------------------------------------------------------------------------------------
@dataclass
class MaskTable:
    """Device-resident description of the modality-mutual mask (what replaces the dense (B,1,L,L) tensor)."""
    rects: Optional[torch.Tensor]          # int32 [B, max_rects, 4]
    col_valid_bits: Optional[torch.Tensor]  # int64 (bit pattern of uint64) [B, ceil(L/64)]
    seq_lens: Optional[torch.Tensor]       # int32 [B]
    L: int
    mask_1d: Optional[torch.Tensor] = None  # int64 [B, L] spliced 1-D mask (for callers that want it)

    @property
    def max_rects(self) -> int:
        return 0 if self.rects is None else int(self.rects.shape[1])

    def token_counts(self, B: int, device) -> torch.Tensor:
        """int32 [B]: index of the last valid column + 1 (the sample's own length when right-padded; L when left-padded).
        NOT `seq_lens`, which is the reference mask's causal extent (the padded length for samples that hold an image)."""
        if self.col_valid_bits is None:
            return torch.full((B,), self.L, dtype=torch.int32, device=device)
        nw = self.col_valid_bits.shape[1]
        shifts = torch.arange(64, device=device, dtype=torch.int64)
        valid = ((self.col_valid_bits[:, :, None] >> shifts) & 1).view(B, nw * 64)[:, : self.L]
        pos = torch.arange(1, self.L + 1, device=device, dtype=torch.int64)
        return (valid * pos).amax(dim=1).to(torch.int32)

    @staticmethod
    def causal(B: int, Lq: int, device) -> "MaskTable":
        return MaskTable(None, None, None, Lq)

    @staticmethod
    def from_host(rects, mask_1d, seq_lens, device) -> "MaskTable":
        """rects: [B][k] of (row_lo,row_hi,col_lo,col_hi); mask_1d: bool/int [B,L]; seq_lens: [B] or None."""
        import numpy as np
        m = np.asarray(mask_1d).astype(bool)
        B, Lq = m.shape
        k = max(1, max(len(r) for r in rects))
        ra = np.zeros((B, k, 4), dtype=np.int32)
        for b, rs in enumerate(rects):
            for i, r in enumerate(rs):
                ra[b, i] = r
        nw = (Lq + 63) // 64
        pad = np.zeros((B, nw * 64), dtype=bool)
        pad[:, :Lq] = m
        bits = np.packbits(pad.reshape(B, nw, 64), axis=-1, bitorder="little").view(np.uint64).reshape(B, nw)
        t = MaskTable(torch.from_numpy(ra).to(device), torch.from_numpy(bits.view(np.int64).copy()).to(device),
                      None if seq_lens is None else torch.tensor(list(seq_lens), dtype=torch.int32, device=device), Lq,
                      torch.from_numpy(m.astype(np.int64)).to(device))
        return t


# ------------------------------------------------------------------------------------------------
def linear(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None,
           act: int = ACT_NONE, res_row_mod: int = 0, out: Optional[torch.Tensor] = None, w2: Optional[torch.Tensor] = None,
           w2_row0: int = 0, n_rows: Optional[int] = None, row_scale: Optional[torch.Tensor] = None,
           row_shift: Optional[torch.Tensor] = None, col_shift: Optional[torch.Tensor] = None,
           stats_out: Optional[RowStats] = None, stats_eps: float = 0.0, preact_out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """y = act(x W^T + bias) [+ residual]; W is an nn.Linear weight [N,K] (K a multiple of 64 for bf16).
    `preact_out` (act SWIGLU, bf16; the training forward): [M, N] buffer that receives the pre-activations g | u; y is then the
    activation of those bf16 values - what linear + swiglu_fwd compute, in one launch.
    Two-segment weight (DecoupledLinear, src/helpers.py:594-603): with `w2`, logical weight row r is w[r] for r < w2_row0 and
    w2[r - w2_row0] beyond; `n_rows` = logical rows (output width, may include padding columns that repeat w2's last row).
    Folded normalisation (see include/aki_mi355x.h, aki_linear_args): `row_scale` / `row_shift` / `col_shift` apply the input's
    RMSNorm / LayerNorm in the epilogue (w then carries the gain), `stats_out` makes this GEMM produce the statistics of ITS
    output for the next one."""
    dev = _dev(x, w, bias, residual, out, w2, row_scale, row_shift, col_shift)
    lib = L.load()
    x2 = _rows2d(x)
    M, K = x2.shape
    N = w.shape[0] if w2 is None else int(n_rows if n_rows is not None else w2_row0 + w2.shape[0])
    if w2 is not None and (w2.shape[1] != K or w2.stride(1) != 1 or w2.stride(0) != w.stride(0) or not 0 < w2_row0 <= w.shape[0]):
        raise AkiError("linear: the second weight segment must have the layout of the first")
    if w.shape[1] != K or w.stride(1) != 1:
        raise AkiError(f"linear: weight {tuple(w.shape)} does not match input K={K}")
    n_out = N // 2 if act == ACT_SWIGLU else N
    if out is None:
        out = torch.empty((*x.shape[:-1], n_out), dtype=x.dtype, device=dev)
    o2 = out.view(-1, out.shape[-1])
    if o2.shape != (M, n_out) or o2.stride(1) != 1:
        raise AkiError("linear: bad output buffer")
    r2 = None
    if residual is not None:
        r2 = _rows2d(residual)
    a = L.LinearArgs(_ptr(x2), _ptr(w), _ptr(bias), _ptr(r2), _ptr(o2), M, N, K, x2.stride(0), w.stride(0), o2.stride(0),
                     0 if r2 is None else r2.stride(0), res_row_mod, act, _dt(x), None, None,
                     _ptr(w2), int(w2_row0) if w2 is not None else 0, int(w2.shape[0]) if w2 is not None else 0)
    a.row_scale, a.row_shift, a.col_shift = _ptr(row_scale), _ptr(row_shift), _ptr(col_shift)
    if stats_out is not None:
        sws = _stats_ws(M, n_out, dev)
        a.stats_rstd, a.stats_mean, a.stats_eps = _ptr(stats_out.rstd), _ptr(stats_out.mean), float(stats_eps)
        a.stats_workspace, a.stats_workspace_bytes = sws.data_ptr(), sws.numel()
    if preact_out is not None:
        p2 = preact_out.view(-1, preact_out.shape[-1])
        if act != ACT_SWIGLU or p2.shape != (M, N) or p2.stride(1) != 1 or p2.dtype != torch.bfloat16 or p2.device != x2.device:
            raise AkiError("linear: preact_out must be a bf16 [M, N] buffer of an act=SWIGLU launch")
        a.preact_out, a.ld_preact = p2.data_ptr(), p2.stride(0)
    if act == ACT_NONE and M <= SPLITK_MAX_M and x.dtype == torch.bfloat16:
        sk = _splitk_ws(M, N, K, dev)
        if sk is not None:
            a.splitk_workspace, a.splitk_workspace_bytes = sk.data_ptr(), sk.numel()
    end = _TAP.begin(("linear", M, N, K, act)) if (_TAP is not None and _TAP.want(("linear", M, N, K, act))) else None
    L.check(lib.aki_linear_fwd(C.byref(a), _stream()), "aki_linear_fwd")
    if end is not None:
        end.record()
    return out


def rmsnorm(x: torch.Tensor, w: torch.Tensor, eps: float) -> torch.Tensor:
    dev = _dev(x, w)
    x2 = _rows2d(x)
    y = torch.empty_like(x2)
    L.check(L.load().aki_rmsnorm_fwd(_ptr(x2), _ptr(w), _ptr(y), x2.shape[0], x2.shape[1], x2.stride(0), y.stride(0),
                                     float(eps), _dt(x), _stream()), "aki_rmsnorm_fwd")
    return y.reshape(x.shape)


def layernorm(x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor], eps: float) -> torch.Tensor:
    dev = _dev(x, w, b)
    x2 = _rows2d(x)
    y = torch.empty_like(x2)
    L.check(L.load().aki_layernorm_fwd(_ptr(x2), _ptr(w), _ptr(b), _ptr(y), x2.shape[0], x2.shape[1], x2.stride(0),
                                       y.stride(0), float(eps), _dt(x), _stream()), "aki_layernorm_fwd")
    return y.reshape(x.shape)


def mma_attn_core(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, table: MaskTable, scale: float,
                  dead_rows: int = DEAD_ROWS_UNIFORM, return_lse: bool = False):
    """q [B,H,L,Dh]; k,v [B,H,cap,Dh] with cap >= L (cap > L: a KV cache whose first L rows are used) -> o [B,L,H*Dh]."""
    dev = _dev(q, k, v)
    B, H, Lq, Dh = q.shape
    cap = k.shape[2]
    if v.shape[2] != cap or cap < Lq:
        raise AkiError("mma_attn_core: k/v capacity mismatch")
    q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
    o = torch.empty((B, Lq, H * Dh), dtype=q.dtype, device=dev)
    lse = torch.empty((B, H, Lq), dtype=torch.float32, device=dev) if return_lse else None
    lib = L.load()
    ws = _ws(lib.aki_mma_attn_core_workspace_bytes(B, H, Lq, Dh, _dt(q)), dev)
    a = L.MmaAttnCoreArgs(_ptr(q), _ptr(k), _ptr(v), _ptr(o), _ptr(lse), _ptr(table.rects), _ptr(table.col_valid_bits),
                          _ptr(table.seq_lens), table.max_rects, B, H, Lq, Dh, float(scale), _dt(q), dead_rows,
                          0 if cap == Lq else cap)
    end = _TAP.begin(("mma_attn_core", B, H, Lq, Dh)) if (_TAP is not None and _TAP.want(("mma_attn_core",))) else None
    L.check(lib.aki_mma_attn_core_fwd(C.byref(a), _ptr(ws), ws.numel(), _stream()), "aki_mma_attn_core_fwd")
    if end is not None:
        end.record()
    return (o, lse) if return_lse else o


def attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, scale: float, return_lse: bool = False):
    """Plain multi-head attention for the vision side.  q [B,Lq,H,Dh], k/v [B,Lk,H,Dh] as (possibly strided) VIEWS of
    the projection outputs (channel stride 1) -> o [B,Lq,H*Dh].  bf16 reads the views in place; f32 (parity path)
    gathers them into contiguous head-major tensors first."""
    dev = _dev(q, k, v)
    B, Lq, H, Dh = q.shape
    Lk = k.shape[1]
    o = torch.empty((B, Lq, H * Dh), dtype=q.dtype, device=dev)
    lib = L.load()
    if q.dtype == torch.float32:
        if Lq != Lk:  # the f32 parity kernel is square: pad the queries (extra rows are discarded)
            qp = torch.zeros((B, Lk, H, Dh), dtype=q.dtype, device=dev)
            qp[:, :Lq] = q
            return attention(qp, k, v, scale)[:, :Lq].contiguous()
        q, k, v = (t_.permute(0, 2, 1, 3).contiguous() for t_ in (q, k, v))     # [B,H,L,Dh]
        st = lambda t_: (t_.stride(0), t_.stride(1), t_.stride(2))
    else:
        for t_ in (q, k, v):
            if t_.stride(3) != 1:
                raise AkiError("attention: channel stride must be 1")
        st = lambda t_: (t_.stride(0), t_.stride(2), t_.stride(1))               # (batch, head, token)
    ws = _ws(B * H * Dh * 4 + 256, dev)
    lse = torch.empty((B, H, Lq), dtype=torch.float32, device=dev) if return_lse else None
    if return_lse and q.dtype != torch.bfloat16:
        raise AkiError("attention: the log-sum-exp output exists on the bf16 path only")
    a = L.AttnArgs(_ptr(q), _ptr(k), _ptr(v), _ptr(o), *st(q), *st(k), *st(v), B, H, Lq, Lk, Dh, float(scale), _dt(q), _ptr(lse))
    end = _TAP.begin(("attention", B, H, Lq, Lk, Dh)) if (_TAP is not None and _TAP.want(("attention",))) else None
    L.check(lib.aki_attn_fwd(C.byref(a), _ptr(ws), ws.numel(), _stream()), "aki_attn_fwd")
    if end is not None:
        end.record()
    return (o, lse) if return_lse else o


def _fused_args(x2, w_qkv, cos, sin, position_ids, o, lse, table, B, H, Lq, Dh, scale, dead_rows, kv_capacity=0, row_scale=None):
    return L.MmaAttnArgs(_ptr(x2), _ptr(w_qkv), _ptr(cos), _ptr(sin), _ptr(position_ids), _ptr(o), _ptr(lse),
                         _ptr(table.rects), _ptr(table.col_valid_bits), _ptr(table.seq_lens), table.max_rects,
                         B, H, Lq, Dh, x2.shape[1], x2.stride(0), w_qkv.stride(0), cos.shape[0], float(scale),
                         _dt(x2), dead_rows, kv_capacity, None, None, _ptr(row_scale))


def mma_attn(x: torch.Tensor, w_qkv: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor, table: MaskTable, num_heads: int,
             scale: Optional[float] = None, position_ids: Optional[torch.Tensor] = None,
             dead_rows: int = DEAD_ROWS_UNIFORM, row_scale: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Fused QKV projection + RoPE + span-driven attention.  x [B,L,d] -> o [B,L,H*Dh] (before o_proj).
    cos/sin: f32 [pos_rows, Dh].  row_scale: folded input RMSNorm (x raw, w_qkv carries the gain)."""
    dev = _dev(x, w_qkv, cos, sin, position_ids, row_scale)
    B, Lq, d = x.shape
    Dh = w_qkv.shape[0] // (3 * num_heads)
    x2 = _rows2d(x)
    cos = cos.to(torch.float32).reshape(-1, Dh).contiguous()
    sin = sin.to(torch.float32).reshape(-1, Dh).contiguous()
    if position_ids is not None:
        position_ids = position_ids.to(torch.int32).expand(B, Lq).contiguous()
    o = torch.empty((B, Lq, num_heads * Dh), dtype=x.dtype, device=dev)
    lib = L.load()
    ws = _ws(lib.aki_mma_attn_workspace_bytes(B, num_heads, Lq, Dh, _dt(x)), dev)
    a = _fused_args(x2, w_qkv, cos, sin, position_ids, o, None, table, B, num_heads, Lq, Dh,
                    scale if scale is not None else Dh ** -0.5, dead_rows, row_scale=row_scale)
    end = _TAP.begin(("mma_attn", B, num_heads, Lq, Dh)) if (_TAP is not None and _TAP.want(("mma_attn",))) else None
    L.check(lib.aki_mma_attn_fwd(C.byref(a), _ptr(ws), ws.numel(), _stream()), "aki_mma_attn_fwd")
    if end is not None:
        end.record()
    return o


def qkv_rope(x: torch.Tensor, w_qkv: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor, num_heads: int,
             position_ids: Optional[torch.Tensor] = None, k_out: Optional[torch.Tensor] = None,
             v_out: Optional[torch.Tensor] = None, row_scale: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """Stage 1 of the fused op: rotated q [B,H,L,Dh], rotated k and v.  With k_out / v_out ([B,H,cap,Dh], cap >= L)
    the keys/values are written straight into a KV cache (prefill)."""
    dev = _dev(x, w_qkv, cos, sin, k_out, v_out)
    B, Lq, d = x.shape
    Dh = w_qkv.shape[0] // (3 * num_heads)
    x2 = _rows2d(x)
    cos = cos.to(torch.float32).reshape(-1, Dh).contiguous()
    sin = sin.to(torch.float32).reshape(-1, Dh).contiguous()
    if position_ids is not None:
        position_ids = position_ids.to(torch.int32).expand(B, Lq).contiguous()
    q = torch.empty((B, num_heads, Lq, Dh), dtype=x.dtype, device=dev)
    if k_out is None:
        k_out = torch.empty((B, num_heads, Lq, Dh), dtype=x.dtype, device=dev)
        v_out = torch.empty((B, num_heads, Lq, Dh), dtype=x.dtype, device=dev)
    cap = k_out.shape[2]
    if not (k_out.is_contiguous() and v_out.is_contiguous()) or v_out.shape != k_out.shape or cap < Lq:
        raise AkiError("qkv_rope: bad KV output buffers")
    a = _fused_args(x2, w_qkv, cos, sin, position_ids, None, None, MaskTable(None, None, None, Lq), B, num_heads, Lq, Dh,
                    Dh ** -0.5, 0, 0 if cap == Lq else cap, row_scale=row_scale)
    ws = _ws(B * Lq * 3 * num_heads * Dh * 4 if x.dtype == torch.float32 else 256, dev)
    L.check(L.load().aki_qkv_rope_fwd(C.byref(a), _ptr(q), _ptr(k_out), _ptr(v_out), _ptr(ws), ws.numel(), _stream()),
            "aki_qkv_rope_fwd")
    return q, k_out, v_out


def rope_append(qkv: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor, pos: torch.Tensor, cache_len: torch.Tensor,
                k_cache: torch.Tensor, v_cache: torch.Tensor, num_heads: int) -> torch.Tensor:
    """Decode step: qkv [B, 3*H*Dh] of the new tokens -> rotated q [B,H,Dh]; k/v appended to the caches in place."""
    dev = _dev(qkv, cos, sin, pos, cache_len, k_cache, v_cache)
    B = qkv.shape[0]
    Dh = qkv.shape[1] // (3 * num_heads)
    q = torch.empty((B, num_heads, Dh), dtype=qkv.dtype, device=dev)
    L.check(L.load().aki_rope_append_fwd(_ptr(qkv.contiguous()), _ptr(cos), _ptr(sin), _ptr(pos), _ptr(cache_len), _ptr(q),
                                         _ptr(k_cache), _ptr(v_cache), B, num_heads, Dh, k_cache.shape[2], _dt(qkv), _stream()),
            "aki_rope_append_fwd")
    return q


def decode_attn_workspace(B: int, H: int, Dh: int, capacity: int, device) -> torch.Tensor:
    """Zero-filled workspace for decode_attn / decode_attn_fused (allocate once per KV cache and keep passing it)."""
    nbytes = int(L.load().aki_decode_attn_workspace_bytes(B, H, Dh, capacity))
    return torch.zeros((nbytes + 3) // 4, dtype=torch.int32, device=device)


def decode_attn(q: torch.Tensor, k_cache: torch.Tensor, v_cache: torch.Tensor, n_keys: torch.Tensor, scale: float,
                col_valid_bits: Optional[torch.Tensor] = None, max_keys: int = 0, ws: Optional[torch.Tensor] = None) -> torch.Tensor:
    """q [B,H,Dh] against the first n_keys[b] rows of the caches [B,H,cap,Dh] -> o [B, H*Dh].
    max_keys: host upper bound of n_keys (0 = capacity); ws: decode_attn_workspace(...) (allocated here if None)."""
    dev = _dev(q, k_cache, v_cache, n_keys, col_valid_bits, ws)
    B, H, Dh = q.shape
    cap = k_cache.shape[2]
    if ws is None:
        ws = decode_attn_workspace(B, H, Dh, cap, dev)
    o = torch.empty((B, H * Dh), dtype=q.dtype, device=dev)
    nw = 0 if col_valid_bits is None else col_valid_bits.shape[1]
    L.check(L.load().aki_decode_attn_fwd(_ptr(q), _ptr(k_cache), _ptr(v_cache), _ptr(o), _ptr(n_keys), _ptr(col_valid_bits), nw,
                                         B, H, Dh, cap, int(max_keys), float(scale), _dt(q), _ptr(ws), ws.numel() * 4, _stream()),
            "aki_decode_attn_fwd")
    return o


def decode_attn_fused(qkv: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor, cache_len: torch.Tensor, k_cache: torch.Tensor,
                      v_cache: torch.Tensor, num_heads: int, scale: float, col_valid_bits: Optional[torch.Tensor] = None,
                      max_keys: int = 0, ws: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Decode step in one launch: qkv [B, 3*H*Dh] of the new tokens -> RoPE at position cache_len[b], k/v appended at
    row cache_len[b], attention over cache_len[b]+1 keys -> o [B, H*Dh].  f32 (parity path) runs the two plain kernels."""
    dev = _dev(qkv, cos, sin, cache_len, k_cache, v_cache, col_valid_bits, ws)
    B = qkv.shape[0]
    cap, Dh = k_cache.shape[2], k_cache.shape[3]
    if qkv.dtype != torch.bfloat16 or Dh != 96:
        q = rope_append(qkv, cos, sin, cache_len, cache_len, k_cache, v_cache, num_heads)
        return decode_attn(q, k_cache, v_cache, cache_len + 1, scale, col_valid_bits, max_keys, ws)
    if ws is None:
        ws = decode_attn_workspace(B, num_heads, Dh, cap, dev)
    o = torch.empty((B, num_heads * Dh), dtype=qkv.dtype, device=dev)
    nw = 0 if col_valid_bits is None else col_valid_bits.shape[1]
    L.check(L.load().aki_decode_attn_fused_fwd(_ptr(qkv.contiguous()), _ptr(cos), _ptr(sin), _ptr(cache_len), _ptr(k_cache),
                                               _ptr(v_cache), _ptr(o), _ptr(col_valid_bits), nw, B, num_heads, Dh, cap,
                                               int(max_keys), float(scale), _dt(qkv), _ptr(ws), ws.numel() * 4, _stream()),
            "aki_decode_attn_fused_fwd")
    return o


_CHAIN_LAST = {}                      # device index -> the torch stream of the last chain launch
_CHAIN_LOCK = threading.Lock()


class DecodeChain:
    """The whole decoder stack of a batch-1 decode step as ONE launch (aki_decode_chain_fwd, decode_chain.hip): a device-resident
    table of per-layer pointers + the workspace with the hand-off vectors and arrival counters.  Built once per (model weights,
    KV cache); `step(h_in)` -> the residual stream after the last layer (pre final norm), bit-identical to the per-layer
    launches.  `check()` reads the sticky error word (a device sync): non-zero means a dependency wait gave up."""

    SUPPORTED = dict(d=3072, F=8192, Dh=96)

    def __init__(self, layers, k_caches, v_caches, H: int, Dh: int, d: int, F: int, capacity: int, scale: float, eps: float, device, w8: bool,
                 batch: int = 1):
        """layers: per layer (w_qkv, w_o, w_gate_up, w_down, norm1, norm2, s_qkv, s_o, s_gate_up, s_down) - scales None for bf16.
        batch 2..8 (bf16): that many sequences per step - the batched chain (16-feature MFMA tiles instead of dot-product rows), bit-identical to
        the per-layer batched launches; k / v caches [batch, H, capacity, Dh]."""
        lib = L.load()
        self.batch = int(batch)
        if self.batch > 1 and (w8 or self.batch > 8 or H != 32):
            raise AkiError("decode chain: batches of 2..8 sequences run on bf16 weights with 32 heads")
        self.n_layers = len(layers)
        rows = []
        for (wq, wo, wg, wd, n1, n2, sq, so, sg, sd), k, v in zip(layers, k_caches, v_caches):
            for t_ in (wq, wo, wg, wd):
                if not t_.is_contiguous():
                    raise AkiError("decode chain: weights must be contiguous [N, K]")
            rows.append([wq.data_ptr(), wo.data_ptr(), wg.data_ptr(), wd.data_ptr(), n1.data_ptr(), n2.data_ptr(), k.data_ptr(), v.data_ptr(),
                         _ptr(sq) or 0, _ptr(so) or 0, _ptr(sg) or 0, _ptr(sd) or 0])
        self.keep = (layers, k_caches, v_caches)                      # the table holds raw pointers: keep the tensors alive
        self.key = tuple(r[0] for r in rows) + tuple(r[6] for r in rows)
        self.table = torch.tensor(rows, dtype=torch.int64).to(device)
        assert C.sizeof(L.DecodeChainLayer) == 12 * 8
        nbytes = int(lib.aki_decode_chain_batch_workspace_bytes(self.n_layers, d, H, F, capacity, self.batch))
        if nbytes == 0:
            raise AkiError("decode chain: bad dimensions")
        self.ws = torch.zeros(nbytes + 256, dtype=torch.uint8, device=device)
        off = (-self.ws.data_ptr()) % 256
        self.ws_ptr = self.ws.data_ptr() + off
        self.ws_bytes = nbytes
        self.err_index = (off + int(lib.aki_decode_chain_batch_error_offset(self.n_layers, H, self.batch))) // 4
        self.h_out = torch.empty((self.batch, d), dtype=torch.bfloat16, device=device)
        self.dims = (H, Dh, d, F, capacity)
        self.scale, self.eps, self.w8 = float(scale), float(eps), bool(w8)

    def step(self, h_in: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor, cache_len: torch.Tensor, col_valid_bits: Optional[torch.Tensor],
             max_keys: int) -> torch.Tensor:
        H, Dh, d, F, cap = self.dims
        if h_in.dtype != torch.bfloat16 or h_in.numel() != self.batch * d or not h_in.is_contiguous():
            raise AkiError("decode chain: h_in must be contiguous bf16 rows [batch, d]")
        if cache_len.numel() != self.batch or (col_valid_bits is not None and col_valid_bits.numel() != self.batch * col_valid_bits.shape[-1]):
            raise AkiError("decode chain: cache_len / col_valid_bits are per sequence")
        if cos.shape[0] < cap or cos.shape[1] != Dh:
            raise AkiError("decode chain: cos/sin tables must cover the cache capacity")
        _dev(h_in, cos, sin, cache_len, col_valid_bits, self.table)
        a = L.DecodeChainArgs(self.table.data_ptr(), _ptr(h_in), _ptr(self.h_out), _ptr(cos), _ptr(sin), _ptr(cache_len), _ptr(col_valid_bits),
                              self.ws_ptr, self.ws_bytes, self.n_layers, 0 if col_valid_bits is None else col_valid_bits.shape[-1], d, H, Dh, F,
                              cap, int(max_keys), self.scale, self.eps, L.AKI_DT_W8A16 if self.w8 else L.AKI_DT_BF16, self.batch)
        end = _TAP.begin(("decode_chain", self.n_layers)) if (_TAP is not None and _TAP.want(("decode_chain",))) else None
        cur = torch.cuda.current_stream()
        with _CHAIN_LOCK:
            # one chain in flight per device (decode_chain.hip, "Progress"): a launch on another stream than the previous chain launch first
            # lets that stream finish.  Same stream = ordered by the stream; nothing to do (the common case: one comparison).
            last = _CHAIN_LAST.get(cur.device_index)
            if last is not None and last.cuda_stream != cur.cuda_stream and not torch.cuda.is_current_stream_capturing():
                last.synchronize()
            _CHAIN_LAST[cur.device_index] = cur
            L.check(L.load().aki_decode_chain_fwd(C.byref(a), cur.cuda_stream), "aki_decode_chain_fwd")
        if end is not None:
            end.record()
        return self.h_out

    def error_code(self) -> int:
        """The sticky error word (synchronises the device): 0, or (layer << 8 | phase) of a wait that gave up."""
        return int(self.ws.view(torch.int32)[self.err_index].item())

    def check(self) -> None:
        code = self.error_code()
        if code:
            raise AkiError(f"decode chain: a dependency wait gave up (layer {code >> 8}, phase {code & 255}); the step's output is invalid")


def chain_replay_on_current_stream() -> None:
    """A hipGraph that holds a decode-chain launch is about to be replayed on the current stream: apply the same one-chain-in-flight-per-device
    rule as DecodeChain.step does for eager launches (a replay is invisible to it otherwise) - let the stream of the previous chain launch
    finish if it is another one, then record this stream."""
    cur = torch.cuda.current_stream()
    with _CHAIN_LOCK:
        last = _CHAIN_LAST.get(cur.device_index)
        if last is not None and last.cuda_stream != cur.cuda_stream:
            last.synchronize()
        _CHAIN_LAST[cur.device_index] = cur


def greedy_pick(logits: torch.Tensor, next_ids: torch.Tensor, pad_token_id: int = 0, eos_ids: Optional[torch.Tensor] = None,
                done: Optional[torch.Tensor] = None, tokens: Optional[torch.Tensor] = None, cache_len: Optional[torch.Tensor] = None,
                start_len: Optional[torch.Tensor] = None, advance: bool = False, done_at: Optional[torch.Tensor] = None,
                embed=None, next_embeds: Optional[torch.Tensor] = None) -> torch.Tensor:
    """The step between two decode steps of a greedy `generate` as one launch (aki_greedy_pick; HF GenerationMixin's greedy branch):
    next_ids[b] = pad if done[b] else argmax(logits[b]); tokens[b, t] = next with t = cache_len[b] + advance - start_len[b]; rows whose
    next is an eos id become done (done_at[b] = t); advance: cache_len += 1.  logits: bf16 [B, V] (row stride >= V); next_ids / tokens /
    eos_ids int64; done uint8; cache_len / start_len / done_at int32.  Capturable: no host value is read.
    embed = (weight [rows, d], additional_weight [extra, d] or None, max_original_id) with next_embeds bf16 [B, d]: the picked token's
    embedding row (DecoupledEmbedding's two tables, src/helpers.py:440-492) is written to next_embeds in the same launch - the input of the next
    decode step."""
    if logits.dtype != torch.bfloat16 or logits.dim() != 2 or logits.stride(1) != 1:
        raise AkiError("greedy_pick takes bf16 logits [B, V] with unit column stride")
    B, V = logits.shape
    for t_, dt in ((next_ids, torch.int64), (tokens, torch.int64), (eos_ids, torch.int64), (done, torch.uint8), (cache_len, torch.int32),
                   (start_len, torch.int32), (done_at, torch.int32)):
        if t_ is not None and (t_.dtype != dt or not t_.is_contiguous() or t_.device != logits.device):
            raise AkiError(f"greedy_pick: a {dt} contiguous tensor on {logits.device} is expected, got {t_.dtype} {tuple(t_.shape)}")
    if tokens is not None and (tokens.dim() != 2 or tokens.shape[0] != B):
        raise AkiError("greedy_pick: tokens is [B, max_new_tokens]")
    common = (_ptr(logits), B, V, logits.stride(0), _ptr(eos_ids), 0 if eos_ids is None else eos_ids.numel(), int(pad_token_id), _ptr(done),
              _ptr(next_ids), _ptr(tokens), 0 if tokens is None else tokens.shape[1], _ptr(cache_len), _ptr(start_len), 1 if advance else 0,
              _ptr(done_at))
    if embed is None:
        L.check(L.load().aki_greedy_pick(*common, _stream()), "aki_greedy_pick")
        return next_ids
    w, extra, max_orig = embed
    d = w.shape[1]
    for t_ in (w, extra, next_embeds):
        if t_ is not None and (t_.dtype != torch.bfloat16 or not t_.is_contiguous() or t_.device != logits.device or t_.shape[-1] != d):
            raise AkiError("greedy_pick: embedding tables and next_embeds are contiguous bf16 [*, d] on the logits' device")
    if next_embeds is None or next_embeds.numel() != B * d:
        raise AkiError("greedy_pick: next_embeds is [B, d]")
    if w.shape[0] <= max_orig:
        raise AkiError("greedy_pick: the embedding table has fewer than max_original_id + 1 rows")
    L.check(L.load().aki_greedy_pick_embed(*common, _ptr(w), _ptr(extra), int(max_orig), 0 if extra is None else extra.shape[0], d,
                                           _ptr(next_embeds), _stream()), "aki_greedy_pick_embed")
    return next_ids


SKINNY_NORM_FUSED = True          # tools (decode_bench.py --norm-launch): False = the RMSNorm of 2-8 row decode GEMMs as a launch of its own


def decode_linear(x: torch.Tensor, w: torch.Tensor, rms_weight: torch.Tensor, eps: float, act: int = ACT_NONE,
                  bias: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None) -> torch.Tensor:
    """y = act(rmsnorm(x; rms_weight, eps) W^T + bias) [+ residual] for the few rows of a decode step: one weight-streaming
    launch when x is bf16 with <= 8 rows (one row: the dot-product GEMV; 2-8 rows: the skinny MFMA GEMM, the norm in its prologue),
    otherwise the norm kernel followed by linear()."""
    x2 = _rows2d(x)
    M, K = x2.shape
    n_tiles = ((w.shape[0] // 2 if act == ACT_SWIGLU else w.shape[0]) + 15) // 16
    # 2-8 rows: every workgroup normalises the rows for itself - the 2004 two-wave workgroups of an lm_head-wide output keep the norm launch
    if (x.dtype != torch.bfloat16 or M > 8 or K > (65536 if M == 1 else 8192) or K % 8 or x2.stride(0) % 8 or w.stride(0) % 8
            or (M > 1 and (n_tiles >= 1536 or K % 256 or not SKINNY_NORM_FUSED))):
        return linear(rmsnorm(x, rms_weight, eps), w, bias=bias, residual=residual, act=act)
    dev = _dev(x, w, rms_weight, bias, residual)
    N = w.shape[0]
    n_out = N // 2 if act == ACT_SWIGLU else N
    out = torch.empty((*x.shape[:-1], n_out), dtype=x.dtype, device=dev)
    o2 = out.view(-1, n_out)
    r2 = None if residual is None else _rows2d(residual)
    a = L.LinearArgs(_ptr(x2), _ptr(w), _ptr(bias), _ptr(r2), _ptr(o2), M, N, K, x2.stride(0), w.stride(0), o2.stride(0),
                     0 if r2 is None else r2.stride(0), 0, act, _dt(x))
    L.check(L.load().aki_decode_linear_fwd(C.byref(a), _ptr(rms_weight), float(eps), _stream()), "aki_decode_linear_fwd")
    return out


# ---- layer loops in one C call (csrc/stack.hip) -------------------------------------------------------------------------
_SCRATCH = {}


def _scratch(nbytes: int, dev) -> torch.Tensor:
    """Uninitialised scratch per (device, stream), 256-byte aligned: launches that use it are stream-ordered."""
    key = (torch.device(dev).index, torch.cuda.current_stream().cuda_stream)
    hit = _SCRATCH.get(key)
    if hit is None or hit.numel() < nbytes + 256:
        hit = _SCRATCH[key] = torch.empty(int(nbytes * 1.25) + 256, dtype=torch.uint8, device=dev)
    return hit


def stack_enabled() -> bool:
    """The one-call layer loops hide the individual launches from the event tap (bench.py brackets single GEMMs): off while one is installed.
    Off under stream capture as well: their scratch buffer is cached across calls and must not come out of a graph's private pool."""
    return _TAP is None and not torch.cuda.is_current_stream_capturing()


def python_must_run_between(layers) -> bool:
    """True when something in Python has to happen inside one of `layers`: a forward (pre-)hook on the layer OR on any of its sub-modules
    (activation capture, adapter wrappers), or an instance-level `forward` override (gradient checkpointing installs one).  The one-call
    layer loops (csrc/stack.hip) never enter Python between layers, so they are only taken when this is False."""
    for ly in layers:
        for m in ly.modules():
            if m._forward_hooks or m._forward_pre_hooks or "forward" in m.__dict__:
                return True
    return False


def params_signature(tensors) -> tuple:
    """(address, in-place version) of every tensor: changes when a parameter is re-allocated (`.to`, `.data = ...`) or written in place
    through torch (`copy_`, an optimizer step).  It signs the OBJECTS it is given: a caller that wants load_state_dict(assign=True) or an
    attribute assignment noticed must pass the module's live parameters (and may add their id()s), or bump a version in its load hooks
    (Phi3Model does the latter, the SigLIP and Perceiver stacks the former).  Writes through raw pointers (this library's trainers) are announced by the
    weight epoch, which callers add themselves.  ~0.25 us per tensor: cheap enough to run before every stacked forward."""
    return tuple((t_.data_ptr(), t_._version) for t_ in tensors)


class LayerTable:
    """A host array of per-layer pointer structs for the stack calls, rebuilt only when a pointer changes."""

    def __init__(self, struct):
        self.struct, self.key, self.arr, self.keep = struct, None, None, None
        self.sig = None                 # the owner's cheap signature of everything the rows were prepared from (see params_signature)

    def get(self, rows):
        """rows: per layer a tuple of tensors / None in the struct's field order."""
        key = tuple(0 if t_ is None else t_.data_ptr() for r in rows for t_ in r)
        if key != self.key:
            n = len(self.struct._fields_)
            arr = (self.struct * len(rows))()
            for i, r in enumerate(rows):
                arr[i] = self.struct(*key[i * n:(i + 1) * n])
            self.key, self.arr, self.keep = key, arr, rows
        return self.arr


def decoder_stack(table_arr, n_layers: int, h: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor, mask: MaskTable, num_heads: int, head_dim: int,
                  inter: int, scale: float, eps: float, position_ids: Optional[torch.Tensor] = None, kv_capacity: int = 0,
                  dead_rows: int = DEAD_ROWS_UNIFORM):
    """All decoder layers of a bf16 inference forward in ONE call (aki_decoder_stack_fwd): h [B, L, d] raw residual stream -> (h_out, 1/rms of
    its rows).  `table_arr`: LayerTable of L.DecoderLayer (gain-folded w_qkv / w_gate_up, w_o, w_down, KV cache tensors or None)."""
    dev = _dev(h, cos, sin)
    B, Lq, d = h.shape
    if h.dtype != torch.bfloat16 or not h.is_contiguous():
        raise AkiError("decoder_stack: h is a contiguous bf16 [B, L, d] tensor")
    lib = L.load()
    cos = cos.to(torch.float32).reshape(-1, head_dim).contiguous()
    sin = sin.to(torch.float32).reshape(-1, head_dim).contiguous()
    pos = None if position_ids is None else position_ids.to(torch.int32).expand(B, Lq).contiguous()
    M = B * Lq
    out = torch.empty_like(h)
    rstd = torch.empty((M,), dtype=torch.float32, device=dev)
    keep = 1 if kv_capacity else 0
    need = int(lib.aki_decoder_stack_workspace_bytes(B, num_heads, Lq, head_dim, d, inter, keep))
    ws = _scratch(need, dev)
    off = (-ws.data_ptr()) % 256
    sws = _stats_ws(M, d, dev)
    sk = None
    if M <= 2048:
        for K_ in (num_heads * head_dim, inter):          # o_proj, down: one buffer (it only ever grows) serves both
            got = _splitk_ws(M, d, K_, dev)
            sk = sk if got is None else got
    a = L.DecoderStackArgs(table_arr, n_layers, _ptr(h), _ptr(out), _ptr(rstd), _ptr(cos), _ptr(sin), _ptr(pos), cos.shape[0], _ptr(mask.rects),
                           _ptr(mask.col_valid_bits), _ptr(mask.seq_lens), mask.max_rects, B, num_heads, Lq, head_dim, d, inter,
                           int(kv_capacity), float(scale), float(eps), dead_rows, ws.data_ptr() + off, ws.numel() - off, sws.data_ptr(), sws.numel(),
                           None if sk is None else sk.data_ptr(), 0 if sk is None else sk.numel())
    L.check(lib.aki_decoder_stack_fwd(C.byref(a), _stream()), "aki_decoder_stack_fwd")
    return out, RowStats(rstd)


def siglip_stack(table_arr, n_layers: int, h: torch.Tensor, fc1_out: torch.Tensor, heads: int, inter: int, act: int, eps: float,
                 scale: float) -> torch.Tensor:
    """All SigLIP encoder layers of a bf16 inference forward in ONE call (aki_siglip_stack_fwd): h [N, L, E] -> h_out.  `fc1_out` [N*L, Ip]:
    the K-padded fc1 output buffer whose pad columns are zero."""
    dev = _dev(h, fc1_out)
    N, Lq, E = h.shape
    if h.dtype != torch.bfloat16 or not h.is_contiguous() or not fc1_out.is_contiguous():
        raise AkiError("siglip_stack: contiguous bf16 tensors")
    lib = L.load()
    M, Ip = N * Lq, fc1_out.shape[-1]
    out = torch.empty_like(h)
    need = int(lib.aki_siglip_stack_workspace_bytes(N, Lq, E, heads))
    ws = _scratch(need, dev)
    off = (-ws.data_ptr()) % 256
    sws = _stats_ws(M, E, dev)
    sk = None
    if M <= 2048:
        for K_ in (E, Ip):                                # out-proj, fc2
            got = _splitk_ws(M, E, K_, dev)
            sk = sk if got is None else got
    a = L.SiglipStackArgs(table_arr, n_layers, _ptr(h), _ptr(out), _ptr(fc1_out), N, Lq, E, heads, inter, Ip, act, float(eps), float(scale), ws.data_ptr() + off,
                          ws.numel() - off, sws.data_ptr(), sws.numel(), None if sk is None else sk.data_ptr(), 0 if sk is None else sk.numel())
    L.check(lib.aki_siglip_stack_fwd(C.byref(a), _stream()), "aki_siglip_stack_fwd")
    return out


def perceiver_stack(table_arr, n_layers: int, x: torch.Tensor, latents: torch.Tensor, norm_w, norm_b, proj_w, proj_b, heads: int, dim_head: int, d_ff: int,
                    scale: float, eps: float) -> torch.Tensor:
    """The Perceiver connector for ONE (sample, image) pair in one call (aki_perceiver_stack_fwd): x [n1, D], latents [n2, D] -> [n2, D_out]."""
    dev = _dev(x, latents, norm_w, norm_b, proj_w, proj_b)
    if x.dtype != torch.bfloat16 or not x.is_contiguous() or not latents.is_contiguous() or x.dim() != 2 or latents.dim() != 2:
        raise AkiError("perceiver_stack: contiguous bf16 [n, D] tensors")
    lib = L.load()
    n1, D = x.shape
    n2 = latents.shape[0]
    D_out = D if proj_w is None else proj_w.shape[0]
    out = torch.empty((n2, D_out), dtype=x.dtype, device=dev)
    ws = _scratch(int(lib.aki_perceiver_stack_workspace_bytes(n1, n2, D, heads, dim_head, d_ff)), dev)
    off = (-ws.data_ptr()) % 256
    a = L.PerceiverStackArgs(table_arr, n_layers, _ptr(x), _ptr(latents), _ptr(norm_w), _ptr(norm_b), _ptr(proj_w), _ptr(proj_b), _ptr(out), n1, n2, D, heads,
                             dim_head, d_ff, D_out, float(scale), float(eps), ws.data_ptr() + off, ws.numel() - off)
    L.check(lib.aki_perceiver_stack_fwd(C.byref(a), _stream()), "aki_perceiver_stack_fwd")
    return out


# ---- fp8 (e4m3) projections: BASELINE configs[4] -------------------------------------------------------------------
def quant_rows_fp8(x: torch.Tensor, rms_weight: Optional[torch.Tensor] = None, eps: float = 0.0):
    """bf16 rows -> (e4m3 bytes [rows, cols] as uint8, f32 scale per row).  With rms_weight the Phi-3 RMSNorm is applied on
    the way (same rounding points as ops.rmsnorm), i.e. this is `quantize(norm(x))` in one HBM pass."""
    dev = _dev(x, rms_weight)
    if x.dtype != torch.bfloat16:
        raise AkiError("quant_rows_fp8 takes bf16 input")
    x2 = _rows2d(x)
    rows, cols = x2.shape
    q = torch.empty((rows, cols), dtype=torch.uint8, device=dev)
    s = torch.empty((rows,), dtype=torch.float32, device=dev)
    L.check(L.load().aki_quant_rows_fp8(_ptr(x2), _ptr(rms_weight), float(eps), _ptr(q), _ptr(s), rows, cols, x2.stride(0),
                                        q.stride(0), _stream()), "aki_quant_rows_fp8")
    return q, s


def linear_fp8(xq: torch.Tensor, xs: torch.Tensor, wq: torch.Tensor, ws: torch.Tensor, bias: Optional[torch.Tensor] = None,
               residual: Optional[torch.Tensor] = None, act: int = ACT_NONE, out_shape=None) -> torch.Tensor:
    """y (bf16) = act((xq * xs) (wq * ws)^T + bias) [+ residual] on the fp8 MFMA path; xq [M,K] / wq [N,K] uint8 (e4m3)."""
    dev = _dev(xq, xs, wq, ws, bias, residual)
    M, K = xq.shape
    N = wq.shape[0]
    n_out = N // 2 if act == ACT_SWIGLU else N
    out = torch.empty((M, n_out), dtype=torch.bfloat16, device=dev)
    r2 = None if residual is None else _rows2d(residual)
    a = L.LinearArgs(_ptr(xq), _ptr(wq), _ptr(bias), _ptr(r2), _ptr(out), M, N, K, xq.stride(0), wq.stride(0), out.stride(0),
                     0 if r2 is None else r2.stride(0), 0, act, L.AKI_DT_FP8_E4M3, _ptr(xs), _ptr(ws))
    end = _TAP.begin(("linear_fp8", M, N, K, act)) if (_TAP is not None and _TAP.want(("linear_fp8", M, N, K, act))) else None
    L.check(L.load().aki_linear_fwd(C.byref(a), _stream()), "aki_linear_fwd[fp8]")
    if end is not None:
        end.record()
    return out if out_shape is None else out.view(*out_shape)


def mma_attn_fp8(xq: torch.Tensor, xs: torch.Tensor, wq: torch.Tensor, ws: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor,
                 table: "MaskTable", B: int, num_heads: int, scale: Optional[float] = None,
                 position_ids: Optional[torch.Tensor] = None, dead_rows: int = DEAD_ROWS_UNIFORM) -> torch.Tensor:
    """Fused MMA op with fp8 QKV projection: xq [B*L, d] e4m3 + scales -> o [B, L, H*Dh] bf16 (RoPE, attention core in bf16)."""
    dev = _dev(xq, xs, wq, ws, cos, sin)
    lib = L.load()
    M, d = xq.shape
    Lq = M // B
    Dh = wq.shape[0] // (3 * num_heads)
    scale = Dh ** -0.5 if scale is None else scale
    o = torch.empty((B, Lq, num_heads * Dh), dtype=torch.bfloat16, device=dev)
    ws_buf = _ws(lib.aki_mma_attn_workspace_bytes(B, num_heads, Lq, Dh, L.AKI_DT_BF16), dev)
    pos = None if position_ids is None else position_ids.to(torch.int32).contiguous()
    a = L.MmaAttnArgs(_ptr(xq), _ptr(wq), _ptr(cos), _ptr(sin), _ptr(pos), _ptr(o), None, _ptr(table.rects),
                      _ptr(table.col_valid_bits), _ptr(table.seq_lens), table.max_rects, B, num_heads, Lq, Dh, d, xq.stride(0),
                      wq.stride(0), cos.shape[0], float(scale), L.AKI_DT_FP8_E4M3, dead_rows, 0, _ptr(xs), _ptr(ws))
    end = _TAP.begin(("mma_attn_fp8", B, num_heads, Lq, Dh)) if (_TAP is not None and _TAP.want(("mma_attn_fp8",))) else None
    L.check(lib.aki_mma_attn_fwd(C.byref(a), _ptr(ws_buf), ws_buf.numel(), _stream()), "aki_mma_attn_fwd[fp8]")
    if end is not None:
        end.record()
    return o


def qkv_rope_fp8(xq: torch.Tensor, xs: torch.Tensor, wq: torch.Tensor, ws: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor, B: int,
                 num_heads: int, position_ids: Optional[torch.Tensor] = None, k_out: Optional[torch.Tensor] = None,
                 v_out: Optional[torch.Tensor] = None):
    """fp8 QKV projection + RoPE -> bf16 q [B,H,L,Dh], k, v (optionally straight into a KV cache, like qkv_rope)."""
    dev = _dev(xq, xs, wq, ws, cos, sin, k_out, v_out)
    M, d = xq.shape
    Lq = M // B
    Dh = wq.shape[0] // (3 * num_heads)
    cos = cos.to(torch.float32).reshape(-1, Dh).contiguous()
    sin = sin.to(torch.float32).reshape(-1, Dh).contiguous()
    pos = None if position_ids is None else position_ids.to(torch.int32).expand(B, Lq).contiguous()
    q = torch.empty((B, num_heads, Lq, Dh), dtype=torch.bfloat16, device=dev)
    if k_out is None:
        k_out = torch.empty((B, num_heads, Lq, Dh), dtype=torch.bfloat16, device=dev)
        v_out = torch.empty((B, num_heads, Lq, Dh), dtype=torch.bfloat16, device=dev)
    cap = k_out.shape[2]
    if not (k_out.is_contiguous() and v_out.is_contiguous()) or v_out.shape != k_out.shape or cap < Lq:
        raise AkiError("qkv_rope_fp8: bad KV output buffers")
    a = L.MmaAttnArgs(_ptr(xq), _ptr(wq), _ptr(cos), _ptr(sin), _ptr(pos), None, None, None, None, None, 0, B, num_heads, Lq, Dh, d,
                      xq.stride(0), wq.stride(0), cos.shape[0], float(Dh ** -0.5), L.AKI_DT_FP8_E4M3, 0, 0 if cap == Lq else cap,
                      _ptr(xs), _ptr(ws))
    wsb = _ws(256, dev)
    L.check(L.load().aki_qkv_rope_fwd(C.byref(a), _ptr(q), _ptr(k_out), _ptr(v_out), _ptr(wsb), wsb.numel(), _stream()),
            "aki_qkv_rope_fwd[fp8]")
    return q, k_out, v_out


def linear_w8(x: torch.Tensor, wq: torch.Tensor, ws: torch.Tensor, bias: Optional[torch.Tensor] = None,
              residual: Optional[torch.Tensor] = None, act: int = ACT_NONE, rms_weight: Optional[torch.Tensor] = None,
              eps: float = 0.0) -> torch.Tensor:
    """Weight-only fp8 (decode in the fp8 configuration): x bf16 [M <= 16, K], wq e4m3 [N, K] with one scale per row; optional fused RMSNorm of
    x (rms_weight).  One row: the dot-product GEMV; 2-16 rows: the skinny MFMA GEMM on e4m3 weights (half the bytes of the bf16 one).
    y bf16 [M, N_out]."""
    dev = _dev(x, wq, ws, bias, residual, rms_weight)
    x2 = _rows2d(x)
    M, K = x2.shape
    if M > 16 or x.dtype != torch.bfloat16:
        raise AkiError("linear_w8 serves up to 16 bf16 rows")
    N = wq.shape[0]
    if M > 1 and rms_weight is not None and (M > 8 or K > 8192 or K % 512 or ((N // 2 if act == ACT_SWIGLU else N) + 15) // 16 >= 1536):
        x2 = _rows2d(rmsnorm(x, rms_weight, eps))      # shapes whose rows the GEMM does not normalise itself (wide outputs, more than eight rows)
        rms_weight = None
    n_out = N // 2 if act == ACT_SWIGLU else N
    out = torch.empty((*x.shape[:-1], n_out), dtype=torch.bfloat16, device=dev)
    o2 = out.view(-1, n_out)
    r2 = None if residual is None else _rows2d(residual)
    a = L.LinearArgs(_ptr(x2), _ptr(wq), _ptr(bias), _ptr(r2), _ptr(o2), M, N, K, x2.stride(0), wq.stride(0), o2.stride(0),
                     0 if r2 is None else r2.stride(0), 0, act, L.AKI_DT_W8A16, None, _ptr(ws))
    lib = L.load()
    if rms_weight is not None:
        L.check(lib.aki_decode_linear_fwd(C.byref(a), _ptr(rms_weight), float(eps), _stream()), "aki_decode_linear_fwd[w8]")
    else:
        L.check(lib.aki_linear_fwd(C.byref(a), _stream()), "aki_linear_fwd[w8]")
    return out


def pad_k(w: torch.Tensor, mult: int = 64) -> torch.Tensor:
    """Zero-pad the K (last) dimension of a weight to a multiple of `mult` (one-time host-side prep)."""
    K = w.shape[-1]
    Kp = (K + mult - 1) // mult * mult
    if Kp == K:
        return w.contiguous()
    out = torch.zeros(*w.shape[:-1], Kp, dtype=w.dtype, device=w.device)
    out[..., :K] = w
    return out


def patch_embed(pixels: torch.Tensor, w_padded: torch.Tensor, bias: Optional[torch.Tensor], pos: Optional[torch.Tensor],
                patch: int) -> torch.Tensor:
    """pixels [N,3,S,S]; w_padded [E,Kp] = pad_k(conv_weight.reshape(E,-1)); pos [G*G,E] -> [N,G*G,E]."""
    dev = _dev(pixels, w_padded, bias, pos)
    N, _, S, _ = pixels.shape
    E, Kp = w_padded.shape
    G = S // patch
    pixels = pixels.contiguous()
    out = torch.empty((N, G * G, E), dtype=pixels.dtype, device=dev)
    lib = L.load()
    ws = _ws(lib.aki_patch_embed_workspace_bytes(N, S, patch, _dt(pixels)), dev)
    L.check(lib.aki_patch_embed_fwd(_ptr(pixels), _ptr(w_padded), _ptr(bias), _ptr(pos), _ptr(out), N, S, patch, E, Kp,
                                    _dt(pixels), _ptr(ws), ws.numel(), _stream()), "aki_patch_embed_fwd")
    return out


def connector_mlp(x: torch.Tensor, ln_w, ln_b, w1, w2, eps: float = 1e-5) -> torch.Tensor:
    """out = x + W2 gelu(W1 LN(x)) - the Perceiver FeedForward block with its residual (src/helpers.py:32-39,194)."""
    dev = _dev(x, ln_w, ln_b, w1, w2)
    x2 = _rows2d(x).contiguous()
    rows, d = x2.shape
    d_inner = w1.shape[0]
    out = torch.empty_like(x2)
    lib = L.load()
    ws = _ws(lib.aki_connector_mlp_workspace_bytes(rows, d, d_inner, _dt(x)), dev)
    L.check(lib.aki_connector_mlp_fwd(_ptr(x2), _ptr(ln_w), _ptr(ln_b), _ptr(w1), _ptr(w2), _ptr(out), rows, d, d_inner,
                                      float(eps), _dt(x), _ptr(ws), ws.numel(), _stream()), "aki_connector_mlp_fwd")
    return out.reshape(x.shape)


def connector_proj(x: torch.Tensor, ln_w, ln_b, w, b, eps: float = 1e-5) -> torch.Tensor:
    """out = Wp LN(x) + bp (src/helpers.py:196-197)."""
    dev = _dev(x, ln_w, ln_b, w, b)
    x2 = _rows2d(x).contiguous()
    rows, d = x2.shape
    d_out = w.shape[0]
    out = torch.empty((rows, d_out), dtype=x.dtype, device=dev)
    lib = L.load()
    ws = _ws(rows * d * x.element_size() + 256, dev)
    L.check(lib.aki_connector_proj_fwd(_ptr(x2), _ptr(ln_w), _ptr(ln_b), _ptr(w), _ptr(b), _ptr(out), rows, d, d_out,
                                       float(eps), _dt(x), _ptr(ws), ws.numel(), _stream()), "aki_connector_proj_fwd")
    return out.reshape(*x.shape[:-1], d_out)


@dataclass
class SplicePlan:
    """The per-sample splice plan (image count, text span, output length ...) on its way to the host: started early by
    `splice_plan_async`, collected by `splice`."""
    plan: torch.Tensor           # int32 [B, AKI_PLAN_STRIDE] on the device
    host: torch.Tensor           # the same, pinned host memory, valid once `ready` has passed
    ready: "torch.cuda.Event"
    lang_x: torch.Tensor         # the int64 ids the plan was made from
    key: tuple


_PLAN_PINNED = {}


def splice_plan_async(lang_x: torch.Tensor, media_token_id: int, assistant_token_id: int, n_vis_tokens: int) -> SplicePlan:
    """The plan depends on the token ids only, while the splice itself needs the vision tokens - the last thing the vision
    side produces.  Sizing the outputs needs the plan on the HOST: called before the vision tower is issued, the plan kernel and
    its 100-byte copy run ahead of ~8 ms of queued GPU work, and `splice` later waits on an event that has long passed instead
    of draining the stream (0.35 ms of idle GPU per forward at the benchmark shape)."""
    dev = _dev(lang_x)
    B, T = lang_x.shape
    ids = lang_x.to(torch.int64).contiguous()
    plan = torch.empty((B, L.AKI_PLAN_STRIDE), dtype=torch.int32, device=dev)
    L.check(L.load().aki_splice_plan(_ptr(ids), B, T, media_token_id, assistant_token_id, n_vis_tokens, _ptr(plan), _stream()),
            "aki_splice_plan")
    ring = _PLAN_PINNED.setdefault((B, torch.device(dev).index), [[], 0])
    if len(ring[0]) < 4:
        ring[0].append(torch.empty((B, L.AKI_PLAN_STRIDE), dtype=torch.int32).pin_memory())
    host = ring[0][ring[1] % len(ring[0])]
    ring[1] += 1
    host.copy_(plan, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record()
    return SplicePlan(plan, host, ev, ids, (lang_x.data_ptr(), lang_x._version, tuple(lang_x.shape), media_token_id, assistant_token_id, n_vis_tokens))


def splice(lang_x: torch.Tensor, attention_mask: Optional[torch.Tensor], labels: Optional[torch.Tensor],
           embed_weight: torch.Tensor, embed_additional: Optional[torch.Tensor], max_original_id: int,
           vision_tokens: torch.Tensor, media_token_id: int, pad_token_id: int, assistant_token_id: int = 32001,
           padding_side: str = "right", max_rects: int = 1, plan: Optional[SplicePlan] = None):
    """Language-stream fusion (src/vlm.py:445-603).  Returns (inputs_embeds, labels_out, MaskTable, plan_host).
    One small device->host copy (the per-sample plan) is needed to size the outputs; `plan` = the result of an earlier
    `splice_plan_async` on the same ids makes that copy free."""
    dev = _dev(lang_x, attention_mask, labels, embed_weight, embed_additional, vision_tokens)
    lib = L.load()
    B, T = lang_x.shape
    if plan is not None and plan.key != (lang_x.data_ptr(), lang_x._version, tuple(lang_x.shape), media_token_id, assistant_token_id,
                                         vision_tokens.shape[2]):
        plan = None                     # made from other ids: plan again
    lang_x = plan.lang_x if plan is not None else lang_x.to(torch.int64).contiguous()
    if attention_mask is not None:
        attention_mask = attention_mask.to(torch.int64).contiguous()
    if labels is not None:
        labels = labels.to(torch.int64).contiguous()
    vision_tokens = vision_tokens.to(embed_weight.dtype).contiguous()
    _, T_img, Nv, d = vision_tokens.shape
    if plan is not None:
        plan.ready.synchronize()
        plan, plan_h = plan.plan, plan.host.clone()
    else:
        plan = torch.empty((B, L.AKI_PLAN_STRIDE), dtype=torch.int32, device=dev)
        L.check(lib.aki_splice_plan(_ptr(lang_x), B, T, media_token_id, assistant_token_id, Nv, _ptr(plan), _stream()),
                "aki_splice_plan")
        plan_h = plan.cpu()
    n_img = plan_h[:, 0]
    if int(n_img.max()) > T_img:
        raise AkiError(f"a sample has {int(n_img.max())} <image> placeholders but vision_x carries only {T_img} images")
    if int(n_img.max()) > max_rects:
        raise AkiError(f"{int(n_img.max())} images in one sample but max_rects={max_rects}")
    L_out = int(plan_h[:, 2].max())
    dt = embed_weight.dtype
    embeds = torch.empty((B, L_out, d), dtype=dt, device=dev)
    labels_out = torch.empty((B, L_out), dtype=torch.int64, device=dev) if labels is not None else None
    mask_1d = torch.empty((B, L_out), dtype=torch.int64, device=dev)
    rects = torch.empty((B, max_rects, 4), dtype=torch.int32, device=dev)
    nw = (L_out + 63) // 64
    bits = torch.empty((B, nw), dtype=torch.int64, device=dev)
    seq_lens = torch.empty((B,), dtype=torch.int32, device=dev)
    a = L.SpliceArgs(_ptr(lang_x), _ptr(attention_mask), _ptr(labels), _ptr(embed_weight), _ptr(embed_additional),
                     _ptr(vision_tokens), _ptr(plan), _ptr(embeds), _ptr(labels_out), _ptr(mask_1d), _ptr(rects),
                     _ptr(bits), _ptr(seq_lens), max_original_id, media_token_id, pad_token_id, B, T, T_img, Nv, d, L_out,
                     max_rects, 1 if padding_side == "left" else 0, _dt(embeds))
    L.check(lib.aki_splice_fwd(C.byref(a), _stream()), "aki_splice_fwd")
    return embeds, labels_out, MaskTable(rects, bits, seq_lens, L_out, mask_1d), plan_h


def mask_to_table(mask: torch.Tensor, max_rects: int = L.AKI_MAX_RECTS, verify: bool = True) -> MaskTable:
    """The reference's LM hand-off type - `attention_mask` (B,1,L,L) 0/1 as returned by `_prepare_inputs_for_forward`
    (src/vlm.py:589-603) - converted on the device into a MaskTable.  The kernels extract a candidate (rectangles from the
    right-of-diagonal row intervals, valid bits from the column-wise OR, seq_lens from the last non-empty row); with
    `verify` the candidate is materialised again and compared with the input bit for bit, so a mask outside the family
    {causal + row-interval rectangles + invalid columns} raises instead of being approximated (one host sync)."""
    dev = _dev(mask)
    if mask.dim() != 4 or mask.shape[1] != 1 or mask.shape[2] != mask.shape[3]:
        raise AkiError(f"mask_to_table: expected (B,1,L,L), got {tuple(mask.shape)}")
    B, _, Lq, _ = mask.shape
    m = mask if mask.dtype == torch.int64 else (mask != 0).to(torch.int64)
    m = m.contiguous()
    lib = L.load()
    rects = torch.empty((B, max_rects, 4), dtype=torch.int32, device=dev)
    bits = torch.empty((B, (Lq + 63) // 64), dtype=torch.int64, device=dev)
    seq_lens = torch.empty((B,), dtype=torch.int32, device=dev)
    status = torch.empty((B,), dtype=torch.int32, device=dev)
    ws = _ws(lib.aki_mma_mask_to_table_workspace_bytes(B, Lq), dev)
    L.check(lib.aki_mma_mask_to_table(_ptr(m), B, Lq, max_rects, _ptr(rects), _ptr(bits), _ptr(seq_lens), _ptr(status),
                                      _ptr(ws), ws.numel(), _stream()), "aki_mma_mask_to_table")
    table = MaskTable(rects, bits, seq_lens, Lq)
    if verify:
        st = status.cpu()
        if int(st.max()) != 0:
            b = int(st.argmax())
            raise AkiError(f"dense attention mask: sample {b} needs {int(st[b])} row groups, more than the {max_rects} rectangles "
                           "the MMA kernels take; pass an ops.MaskTable or a causal/padding mask")
        if not torch.equal(mask_dense(table, B), (m != 0).to(torch.int64)):
            raise AkiError("dense attention mask is not of the modality-mutual family (causal triangle + per-row column "
                           "intervals + invalid columns); the MI355X path refuses it rather than approximating it")
    return table


def mask_dense(table: MaskTable, B: int) -> torch.Tensor:
    """The reference's (B,1,L,L) int64 0/1 mask, materialised from the table (bit-exact)."""
    dev = table.rects.device if table.rects is not None else table.col_valid_bits.device
    out = torch.empty((B, 1, table.L, table.L), dtype=torch.int64, device=dev)
    L.check(L.load().aki_mma_mask_dense(_ptr(table.rects), table.max_rects, _ptr(table.col_valid_bits), _ptr(table.seq_lens),
                                        B, table.L, _ptr(out), _stream()), "aki_mma_mask_dense")
    return out
