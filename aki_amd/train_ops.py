"""Training-step front end: raw wrappers of the backward / optimizer entry points of the C ABI and the
``torch.autograd.Function`` classes that pair every forward HIP op of the AKI path with its HIP backward.

What this replaces in the reference: ``loss.backward()`` = torch autograd over the eager forward under bf16 autocast
(train/train_utils.py:242-252).  Autograd itself stays (graph bookkeeping is plumbing); every gradient is computed by a
kernel of libaki_mi355x.so.  bf16 only - the weights the forward reads are the bf16 image of the fp32 master weights the
optimizer owns (aki_amd/trainer.py), which is what autocast does to the reference's fp32 parameters on every step.

Weight gradients: if a parameter carries ``_aki_grad`` (a bf16 view into the trainer's flat gradient buffer) the wgrad
GEMM writes straight into it and autograd gets ``None`` for that parameter - no second copy of the gradient ever exists.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import weakref

import torch

from . import _lib as L
from . import ops
from ._lib import AkiError
from .ops import _dev, _dt, _ptr, _stream, _ws, _rows2d

_BF16 = L.AKI_DT_BF16


def _pad64(n: int) -> int:
    return (n + 63) // 64 * 64


def _need_bf16(*ts):
    for t in ts:
        if t is not None and t.dtype != torch.bfloat16:
            raise AkiError("the training kernels are bf16 (fp32 master weights live in the optimizer)")


# ---- raw wrappers ----------------------------------------------------------------------------------------------
def transpose(x: torch.Tensor, pad_to: int = 64, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x [R, C] -> [C, pad(R)] with the padding columns zero (operand of the dgrad / wgrad GEMMs)."""
    _need_bf16(x)
    dev = _dev(x)
    R, Cc = x.shape
    Rp = (R + pad_to - 1) // pad_to * pad_to
    if out is None:
        out = torch.empty((Cc, Rp), dtype=x.dtype, device=dev)
    L.check(L.load().aki_transpose(_ptr(x), _ptr(out), R, Cc, x.stride(0), out.stride(0), Rp, _BF16, _stream()), "aki_transpose")
    return out


def gemm_tn(a: torch.Tensor, b: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """c [I, J] = a[Kc, I]^T @ b[Kc, J]  (bf16, f32 accumulate): a weight gradient dW = dY^T X on the operands as they lie - both tiles are staged
    row-major and read transposed from LDS (aki_gemm_tn), no transpose pass.  I, J and the row strides multiples of 8."""
    _need_bf16(a, b)
    dev = _dev(a, b, out)
    Kc, I = a.shape
    J = b.shape[1]
    if b.shape[0] != Kc or a.stride(1) != 1 or b.stride(1) != 1:
        raise AkiError("gemm_tn: operands must be [Kc, I] and [Kc, J] with unit column stride")
    if out is None:
        out = torch.empty((I, J), dtype=a.dtype, device=dev)
    if out.shape != (I, J) or out.stride(1) != 1:
        raise AkiError("gemm_tn: bad output buffer")
    L.check(L.load().aki_gemm_tn(_ptr(a), _ptr(b), _ptr(out), Kc, I, J, a.stride(0), b.stride(0), out.stride(0), _BF16, _stream()), "aki_gemm_tn")
    return out


USE_GEMM_TN = True    # tools/train_step_ab_tn.py flips it to time the transposes + forward-GEMM form on the same box


def _tn_ok(a: torch.Tensor, b: torch.Tensor) -> bool:
    """Every condition of gemm_tn_launch (gemm_tn_bf16.hip), so that an operand it would refuse takes the transposes +
    forward-GEMM form instead of raising inside backward - including its 32-bit buffer offsets: ((Kc-1)*ld + cols)*2 < 4 GiB
    (e.g. > 131k token rows against the 16384-wide gate_up output)."""
    span = lambda t: ((t.shape[0] - 1) * t.stride(0) + t.shape[1]) * 2
    return USE_GEMM_TN and (a.shape[1] % 8 == 0 and b.shape[1] % 8 == 0 and a.shape[1] >= 8 and b.shape[1] >= 8 and a.stride(1) == 1 and b.stride(1) == 1
            and a.stride(0) % 8 == 0 and b.stride(0) % 8 == 0 and a.data_ptr() % 16 == 0 and b.data_ptr() % 16 == 0
            and span(a) < (1 << 32) and span(b) < (1 << 32))


def norm_bwd(rms: bool, x: torch.Tensor, w: torch.Tensor, dy: torch.Tensor, eps: float, need_db: bool = False,
             dw_out: Optional[torch.Tensor] = None, db_out: Optional[torch.Tensor] = None, dres: Optional[torch.Tensor] = None):
    """dx, dw, db of RMSNorm / LayerNorm.  dw_out / db_out: write the weight gradients there (e.g. a view of the trainer's
    flat gradient buffer) instead of into fresh tensors."""
    _need_bf16(x, w, dy)
    dev = _dev(x, w, dy)
    x2, dy2 = _rows2d(x), _rows2d(dy)
    rows, cols = x2.shape
    dx = torch.empty_like(x2)
    dw = dw_out if dw_out is not None else torch.empty((cols,), dtype=x.dtype, device=dev)
    db = (db_out if db_out is not None else torch.empty((cols,), dtype=x.dtype, device=dev)) if need_db else None
    lib = L.load()
    ws = _ws(lib.aki_norm_bwd_workspace_bytes(cols), dev)
    r2 = None if dres is None else _rows2d(dres)
    L.check(lib.aki_norm_bwd(1 if rms else 0, _ptr(x2), _ptr(w), _ptr(dy2), _ptr(r2), _ptr(dx), _ptr(dw), _ptr(db), rows, cols,
                             x2.stride(0), dy2.stride(0), 0 if r2 is None else r2.stride(0), dx.stride(0), float(eps), 0, _BF16,
                             _ptr(ws), ws.numel(), _stream()), "aki_norm_bwd")
    return dx.view(x.shape), dw, db


def colsum(x: torch.Tensor) -> torch.Tensor:
    _need_bf16(x)
    dev = _dev(x)
    x2 = _rows2d(x)
    rows, cols = x2.shape
    out = torch.empty((cols,), dtype=x.dtype, device=dev)
    lib = L.load()
    ws = _ws(lib.aki_colsum_workspace_bytes(cols), dev)
    L.check(lib.aki_colsum(_ptr(x2), _ptr(out), rows, cols, x2.stride(0), 0, _BF16, _ptr(ws), ws.numel(), _stream()), "aki_colsum")
    return out


def swiglu_fwd(gu: torch.Tensor) -> torch.Tensor:
    _need_bf16(gu)
    g2 = _rows2d(gu)
    rows, F2 = g2.shape
    a = torch.empty((*gu.shape[:-1], F2 // 2), dtype=gu.dtype, device=_dev(gu))
    a2 = a.view(-1, F2 // 2)
    L.check(L.load().aki_swiglu_fwd(_ptr(g2), _ptr(a2), rows, F2 // 2, g2.stride(0), a2.stride(0), _BF16, _stream()), "aki_swiglu_fwd")
    return a


def swiglu_bwd(gu: torch.Tensor, da: torch.Tensor) -> torch.Tensor:
    _need_bf16(gu, da)
    g2, d2 = _rows2d(gu), _rows2d(da)
    rows, F2 = g2.shape
    dgu = torch.empty_like(g2)
    L.check(L.load().aki_swiglu_bwd(_ptr(g2), _ptr(d2), _ptr(dgu), rows, F2 // 2, g2.stride(0), d2.stride(0), dgu.stride(0), _BF16,
                                    _stream()), "aki_swiglu_bwd")
    return dgu.view(gu.shape)


def gelu_fwd(x: torch.Tensor) -> torch.Tensor:
    _need_bf16(x)
    xc = x.contiguous()
    y = torch.empty_like(xc)
    L.check(L.load().aki_gelu_fwd(_ptr(xc), _ptr(y), xc.numel(), _BF16, _stream()), "aki_gelu_fwd")
    return y


def gelu_bwd(x: torch.Tensor, dy: torch.Tensor) -> torch.Tensor:
    _need_bf16(x, dy)
    xc, dc = x.contiguous(), dy.contiguous()
    dx = torch.empty_like(xc)
    L.check(L.load().aki_gelu_bwd(_ptr(xc), _ptr(dc), _ptr(dx), xc.numel(), _BF16, _stream()), "aki_gelu_bwd")
    return dx


def rope_bwd_merge(dq, dk, dv, cos, sin, position_ids=None) -> torch.Tensor:
    _need_bf16(dq, dk, dv)
    B, H, Lq, Dh = dq.shape
    out = torch.empty((B, Lq, 3 * H * Dh), dtype=dq.dtype, device=_dev(dq, dk, dv, cos, sin))
    pos = None if position_ids is None else position_ids.to(torch.int32).contiguous()
    L.check(L.load().aki_rope_bwd_merge(_ptr(dq.contiguous()), _ptr(dk.contiguous()), _ptr(dv.contiguous()), _ptr(cos), _ptr(sin),
                                        _ptr(pos), _ptr(out), B, H, Lq, Dh, _BF16, _stream()), "aki_rope_bwd_merge")
    return out


def ce_loss(logits: torch.Tensor, labels: torch.Tensor, n_cols: int, gscale: float = 1.0, want_grad: bool = True):
    """HF shifted CE over logits [B, L, ld>=n_cols] (first n_cols columns are real).  Returns (loss scalar f32 tensor,
    n_valid int32 device tensor); with want_grad the logits buffer is OVERWRITTEN by d(loss)/d(logits) * gscale."""
    _need_bf16(logits)
    dev = _dev(logits, labels)
    B, Lq, ld = logits.shape
    if logits.stride(2) != 1 or logits.stride(1) != ld or labels.shape != (B, Lq):
        raise AkiError("ce_loss: logits must be [B, L, ld] with dense rows; labels [B, L]")
    lab = labels.to(torch.int64).contiguous()
    rows = torch.empty((B * Lq,), dtype=torch.float32, device=dev)
    nv = torch.empty((1,), dtype=torch.int32, device=dev)
    L.check(L.load().aki_ce_loss_fwd_bwd(_ptr(logits), _ptr(lab), _ptr(nv), _ptr(rows), _ptr(logits) if want_grad else None, B, Lq,
                                         n_cols, logits.stride(1), logits.stride(1), float(gscale), _BF16, _stream()), "aki_ce_loss_fwd_bwd")
    return rows.sum() / nv.clamp(min=1).to(torch.float32)[0], nv


def ce_rows(logits: torch.Tensor, targets: torch.Tensor, n_valid: torch.Tensor, n_cols: int, want_grad: bool) -> torch.Tensor:
    """Cross-entropy of a CHUNK of rows: logits [rows, ld >= n_cols] bf16, targets [rows] int64 (already shifted; < 0 = ignored),
    n_valid: device int32 [1] = scored rows of the whole batch.  Returns loss_rows f32 [rows] (their sum / n_valid is the chunk's
    share of the mean loss); with want_grad the logits buffer is overwritten by d(mean loss)/d(logits)."""
    _need_bf16(logits)
    _dev(logits, targets, n_valid)
    rows = torch.empty((logits.shape[0],), dtype=torch.float32, device=logits.device)
    L.check(L.load().aki_ce_rows_fwd_bwd(_ptr(logits), _ptr(targets), _ptr(n_valid), _ptr(rows), _ptr(logits) if want_grad else None,
                                         logits.shape[0], n_cols, logits.stride(0), logits.stride(0), 1.0, _BF16, _stream()),
            "aki_ce_rows_fwd_bwd")
    return rows


def attn_bwd(q, k, v, o, d_o, lse, table: Optional[ops.MaskTable], scale: float):
    """q,k,v [B,H,L,Dh]; o, d_o [B,Lq,H*Dh]; lse [B,H,Lq] -> dq, dk, dv.  table=None: plain (non-causal) attention."""
    _need_bf16(q, k, v, o, d_o)
    dev = _dev(q, k, v, o, d_o, lse)
    B, H, Lq, Dh = q.shape
    Lk = k.shape[2]
    q, k, v, o, d_o = (t.contiguous() for t in (q, k, v, o, d_o))
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    lib = L.load()
    ws = _ws(lib.aki_attn_bwd_workspace_bytes(B, H, Lq), dev)
    masked = table is not None
    a = L.AttnBwdArgs(_ptr(q), _ptr(k), _ptr(v), _ptr(o), _ptr(d_o), _ptr(lse), _ptr(dq), _ptr(dk), _ptr(dv),
                      _ptr(table.rects) if masked else None, table.max_rects if masked else 0,
                      _ptr(table.col_valid_bits) if masked else None, _ptr(table.seq_lens) if masked else None,
                      1 if masked else 0, B, H, Lq, Lk, Dh, float(scale), _BF16)
    L.check(lib.aki_attn_bwd(C.byref(a), _ptr(ws), ws.numel(), _stream()), "aki_attn_bwd")
    return dq, dk, dv


def grad_sqnorm(g: torch.Tensor, out: torch.Tensor, accumulate: bool = False) -> None:
    """*out (+)= sum g^2; g bf16, or f32 (the fp32 gradient exchange)."""
    if g.dtype != torch.float32:
        _need_bf16(g)
    lib = L.load()
    ws = _ws(lib.aki_grad_sqnorm_workspace_bytes(), _dev(g, out))
    L.check(lib.aki_grad_sqnorm(_ptr(g), g.numel(), _ptr(out), 1 if accumulate else 0, L.AKI_DT_F32 if g.dtype == torch.float32 else _BF16,
                                _ptr(ws), ws.numel(), _stream()), "aki_grad_sqnorm")


def adamw_step(p32, m, v, g, w16, sqnorm, max_norm, gscale, lr, beta1, beta2, eps, wd, step) -> None:
    """g: bf16 gradients, or f32 (aki_adamw_step_g32)."""
    _dev(p32, m, v, g, w16, sqnorm)
    fn = L.load().aki_adamw_step_g32 if g.dtype == torch.float32 else L.load().aki_adamw_step
    L.check(fn(_ptr(p32), _ptr(m), _ptr(v), _ptr(g), _ptr(w16), p32.numel(), _ptr(sqnorm), float(max_norm), float(gscale), float(lr),
               float(beta1), float(beta2), float(eps), float(wd), int(step), _stream()), "aki_adamw_step")


def adamw_step_t(p32, m, v, g, w16, wT, N, K, sqnorm, max_norm, gscale, lr, beta1, beta2, eps, wd, step) -> None:
    """adamw_step for one 2-D weight [N, K] (flat views of N*K elements) that also writes W^T into wT [K, pad64(N)]."""
    _dev(p32, m, v, g, w16, wT, sqnorm)
    L.check(L.load().aki_adamw_step_t(_ptr(p32), _ptr(m), _ptr(v), _ptr(g), _ptr(w16), _ptr(wT), int(N), int(K), int(wT.shape[1]), _ptr(sqnorm),
                                      float(max_norm), float(gscale), float(lr), float(beta1), float(beta2), float(eps), float(wd), int(step),
                                      L.AKI_DT_F32 if g.dtype == torch.float32 else _BF16, _stream()), "aki_adamw_step_t")


def register_weight_t(w: torch.Tensor, wT: torch.Tensor) -> None:
    """Hand the cache a transposed copy somebody else made of the CURRENT weights (the trainer's AdamW pass): _weight_t(w) returns it
    until the next epoch bump."""
    _WT[(w.data_ptr(), tuple(w.shape), w._version, _EPOCH)] = (weakref.ref(w), wT)


# ---- transposed-weight cache -------------------------------------------------------------------------------------
_EPOCH = 0            # bumped by the trainer after every optimizer step (the kernels write weights through raw pointers)
_WT = {}
CACHE_WT = True       # kill switch for every cached weight transform (tools flip it); sharded parameters opt out one by one, below


def sharded(w) -> bool:
    """Does this weight belong to a parameter-sharded unit (AkiShardedTrainer)?  A cached transform of it (transposed copy,
    gain-folded copy) would keep full-size what sharding releases after every use - such weights are never cached.  Per weight:
    until round 4 constructing a sharded trainer switched the caches off for every model of the process, for good."""
    return getattr(w, "_aki_unit", None) is not None


def bump_weight_epoch() -> None:
    global _EPOCH
    _EPOCH += 1
    _WT.clear()


def _weight_t(w: torch.Tensor) -> torch.Tensor:
    """W [N,K] -> W^T [K, pad64(N)] (zero padded), cached until the weights change: the trainer bumps the epoch after every
    optimizer step (its kernels write through raw pointers), in-place torch updates show up in `_version`.  Temporaries (the
    lm_head's concatenated weight is a new tensor every forward) would pile up without a trainer, hence the size cap."""
    if not CACHE_WT or sharded(w) or not (isinstance(w, torch.nn.Parameter) or hasattr(w, "_aki_grad") or getattr(w, "_aki_cacheable", False)):
        # a temporary (slice / concatenation built inside a forward): its (data_ptr, _version) says nothing about its content -
        # the allocator recycles the address and a fresh tensor is always version 0 - so it is transposed every time
        return transpose(w if w.stride(1) == 1 else w.contiguous())
    key = (w.data_ptr(), tuple(w.shape), w._version, _EPOCH)
    hit = _WT.get(key)
    # The key alone cannot tell two weights apart over time: a deleted model's addresses are recycled for the next one's parameters - same
    # shape, version 0, same epoch - and the entry would hand out ANOTHER weight's transpose (round 5: two models built in one process
    # gave garbage input gradients for the second).  An entry therefore remembers WHICH tensor object it was made from.
    if hit is not None and hit[0]() is w:
        return hit[1]
    if len(_WT) >= 320:                      # > every 2-D weight of AKI-4B (32 x 4 + connector + heads)
        _WT.clear()
    t = transpose(w if w.stride(1) == 1 else w.contiguous())
    _WT[key] = (weakref.ref(w), t)
    return t


def _fresh_target(param) -> Optional[torch.Tensor]:
    """The flat-buffer view a kernel may write this parameter's gradient into directly (None: no trainer, or the view
    already holds a gradient of this accumulation window and has to be added to)."""
    if param is None or getattr(param, "_aki_grad_live", False):
        return None
    return getattr(param, "_aki_grad", None)


def _deliver(param: torch.Tensor, grad_writer):
    """Give a weight gradient to its owner: write into the trainer's flat buffer when there is one (returns None for
    autograd), else return a fresh tensor."""
    tgt = getattr(param, "_aki_grad", None)
    if tgt is not None:
        if getattr(param, "_aki_grad_live", False):        # later micro-batch of an accumulation window (or a second use): add
            tgt += grad_writer(None)
        else:
            grad_writer(tgt)
            param._aki_grad_live = True
        hook = getattr(param, "_aki_grad_hook", None)
        if hook is not None:
            hook(param)                                    # lets the reducer launch the bucket this gradient completes
        return None
    return grad_writer(None)


def _wgrad(param, dy2: torch.Tensor, x2: torch.Tensor):
    """dW = dY^T X delivered to the parameter's owner: on the operands as they lie (aki_gemm_tn) when their layout allows, else through two
    transposes and the forward GEMM."""
    if _tn_ok(dy2, x2):
        def write(out):
            if out is None or (out.stride(0) % 4 == 0 and out.data_ptr() % 8 == 0):
                return gemm_tn(dy2, x2, out=out)
            return out.copy_(gemm_tn(dy2, x2))
        return _deliver(param, write)
    dyT, xT = transpose(dy2), transpose(x2)       # [N, Mp], [K, Mp]
    return _deliver(param, lambda out: ops.linear(dyT, xT, out=out))


# ---- autograd Functions ---------------------------------------------------------------------------------------------
class LinearFn(torch.autograd.Function):
    """y = x W^T + b [+ residual]   (HIP MFMA GEMM both ways: dX = dY W, dW = dY^T X, db = colsum dY)."""

    @staticmethod
    def forward(ctx, x, w, bias, residual):
        _need_bf16(x, w)
        y = ops.linear(x, w, bias=bias, residual=residual)
        # Weights are NOT handed to save_for_backward: the backward reads the parameter as it is THEN (ctx.w_ref).  With
        # sharded parameters (trainer.AkiShardedTrainer) its storage is released after the forward and gathered again -
        # in place - before the backward, which autograd's saved-tensor version check would refuse.
        ctx.save_for_backward(x)
        ctx.has_bias, ctx.has_res = bias is not None, residual is not None
        ctx.bias_ref = bias
        ctx.w_ref = w
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,), w = ctx.saved_tensors, ctx.w_ref
        N, K = w.shape
        dy2 = _rows2d(dy)
        M = dy2.shape[0]
        dx = dw = db = None
        if N % 64:                                         # K dimension of the dgrad GEMM must be a multiple of 64
            pad = torch.zeros((M, _pad64(N)), dtype=dy2.dtype, device=dy2.device)
            pad[:, :N] = dy2
            dy2p = pad
        else:
            dy2p = dy2
        if ctx.needs_input_grad[0]:
            dx = ops.linear(dy2p, _weight_t(w)).view(x.shape)
        if ctx.needs_input_grad[1]:
            x2 = _rows2d(x)
            dw = _wgrad(ctx.w_ref, dy2, x2)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = _deliver(ctx.bias_ref, lambda out: colsum(dy2) if out is None else out.copy_(colsum(dy2)))
        return dx, dw, db, (dy if ctx.has_res else None)


def linear(x, w, bias=None, residual=None):
    return LinearFn.apply(x, w, bias, residual)


class NormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, eps, rms):
        y = ops.rmsnorm(x, w, eps) if rms else ops.layernorm(x, w, b, eps)
        ctx.save_for_backward(x)
        ctx.eps, ctx.rms, ctx.w_ref, ctx.b_ref = eps, rms, w, b
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,), w = ctx.saved_tensors, ctx.w_ref
        need_db = not ctx.rms and ctx.b_ref is not None
        wt = _fresh_target(ctx.w_ref) if ctx.needs_input_grad[1] else None
        bt = _fresh_target(ctx.b_ref) if (need_db and ctx.needs_input_grad[2]) else None
        dx, dw, db = norm_bwd(ctx.rms, x, w, dy, ctx.eps, need_db=need_db, dw_out=wt, db_out=bt)   # written in place when possible
        gw = _deliver(ctx.w_ref, lambda out: dw if (out is None or out is wt) else out.copy_(dw)) if ctx.needs_input_grad[1] else None
        gb = None
        if db is not None and ctx.needs_input_grad[2]:
            gb = _deliver(ctx.b_ref, lambda out: db if (out is None or out is bt) else out.copy_(db))
        return dx, gw, gb, None, None


class NormResidualFn(torch.autograd.Function):
    """Pre-norm block entry: (y, h) = (rmsnorm(h), h).  The second output is the residual stream handed to the block's
    output projection; having it leave through this node means the backward receives BOTH gradients (through the norm and
    through the residual) and the kernel sums them - no separate elementwise add over [tokens, d] per block."""

    @staticmethod
    def forward(ctx, x, w, eps):
        y = ops.rmsnorm(x, w, eps)
        ctx.save_for_backward(x)
        ctx.eps, ctx.w_ref = eps, w
        return y, x.view_as(x)

    @staticmethod
    def backward(ctx, dy, dres):
        (x,), w = ctx.saved_tensors, ctx.w_ref
        wt = _fresh_target(ctx.w_ref) if ctx.needs_input_grad[1] else None
        dx, dw, _ = norm_bwd(True, x, w, dy, ctx.eps, dw_out=wt, dres=dres)
        gw = _deliver(ctx.w_ref, lambda out: dw if (out is None or out is wt) else out.copy_(dw)) if ctx.needs_input_grad[1] else None
        return dx, gw, None


def rmsnorm_residual(x, w, eps):
    return NormResidualFn.apply(x, w, eps)


def rmsnorm(x, w, eps):
    return NormFn.apply(x, w, None, eps, True)


def layernorm(x, w, b, eps):
    return NormFn.apply(x, w, b, eps, False)


class SwigluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, gu):
        ctx.save_for_backward(gu)
        return swiglu_fwd(gu)

    @staticmethod
    def backward(ctx, da):
        (gu,) = ctx.saved_tensors
        return swiglu_bwd(gu, da)


class GateUpSwigluFn(torch.autograd.Function):
    """a = silu(g) * u with [g | u] = x W^T in ONE launch (the GEMM's SwiGLU epilogue also leaves the bf16 pre-activations for this
    backward: aki_linear_args.preact_out) - the numbers LinearFn + SwigluFn produce, without the pass that re-read [M, 2F] to apply
    the activation (HF:phi3/modeling_phi3.py:49-64)."""

    @staticmethod
    def forward(ctx, x, w):
        _need_bf16(x, w)
        gu = torch.empty((*x.shape[:-1], w.shape[0]), dtype=x.dtype, device=x.device)
        a = ops.linear(x, w, act=ops.ACT_SWIGLU, preact_out=gu)
        ctx.save_for_backward(x, gu)
        ctx.w_ref = w
        return a

    @staticmethod
    def backward(ctx, da):
        (x, gu), w = ctx.saved_tensors, ctx.w_ref
        dgu = swiglu_bwd(gu, da.contiguous())
        d2 = _rows2d(dgu)
        N = w.shape[0]
        if N % 64:                                         # K dimension of the dgrad GEMM must be a multiple of 64 (as LinearFn)
            d2p = torch.zeros((d2.shape[0], _pad64(N)), dtype=d2.dtype, device=d2.device)
            d2p[:, :N] = d2
        else:
            d2p = d2
        dx = ops.linear(d2p, _weight_t(w)).view(x.shape) if ctx.needs_input_grad[0] else None
        dw = _wgrad(ctx.w_ref, d2, _rows2d(x)) if ctx.needs_input_grad[1] else None
        return dx, dw


def gate_up_swiglu(x, w):
    return GateUpSwigluFn.apply(x, w)


class GeluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return gelu_fwd(x)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return gelu_bwd(x, dy)


class QkvRopeFn(torch.autograd.Function):
    """Fused qkv_proj + RoPE + head split (MFMA GEMM epilogue) -> q, k, v [B,H,L,Dh]; backward = inverse rotation +
    merge, then the two GEMMs of the projection."""

    @staticmethod
    def forward(ctx, x, w, cos, sin, num_heads, position_ids):
        q, k, v = ops.qkv_rope(x, w, cos, sin, num_heads, position_ids)
        ctx.save_for_backward(x, cos, sin)
        ctx.pos, ctx.w_ref = position_ids, w
        return q, k, v

    @staticmethod
    def backward(ctx, dq, dk, dv):
        (x, cos, sin), w = ctx.saved_tensors, ctx.w_ref
        dqkv = rope_bwd_merge(dq, dk, dv, cos, sin, ctx.pos)          # [B, L, 3*H*Dh]
        d2 = dqkv.view(-1, dqkv.shape[-1])
        dx = ops.linear(d2, _weight_t(w)).view(x.shape) if ctx.needs_input_grad[0] else None
        dw = None
        if ctx.needs_input_grad[1]:
            dw = _wgrad(ctx.w_ref, d2, _rows2d(x))
        return dx, dw, None, None, None, None


class MmaAttnCoreFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, table, scale):
        o, lse = ops.mma_attn_core(q, k, v, table, scale, return_lse=True)
        ctx.save_for_backward(q, k, v, o, lse)
        ctx.table, ctx.scale = table, scale
        return o

    @staticmethod
    def backward(ctx, d_o):
        q, k, v, o, lse = ctx.saved_tensors
        dq, dk, dv = attn_bwd(q, k, v, o, d_o, lse, ctx.table, ctx.scale)
        return dq, dk, dv, None, None


class PlainAttnFn(torch.autograd.Function):
    """Non-causal attention of the Perceiver (q [B,Lq,H,Dh], k/v [B,Lk,H,Dh] strided views -> o [B,Lq,H*Dh])."""

    @staticmethod
    def forward(ctx, q, k, v, scale):
        o, lse = ops.attention(q, k, v, scale, return_lse=True)
        ctx.save_for_backward(q, k, v, o, lse)
        ctx.scale = scale
        return o

    @staticmethod
    def backward(ctx, d_o):
        q, k, v, o, lse = ctx.saved_tensors
        qh, kh, vh = (t.permute(0, 2, 1, 3).contiguous() for t in (q, k, v))
        dq, dk, dv = attn_bwd(qh, kh, vh, o, d_o, lse, None, ctx.scale)
        return dq.permute(0, 2, 1, 3), dk.permute(0, 2, 1, 3), dv.permute(0, 2, 1, 3), None


class CELossFn(torch.autograd.Function):
    """Shifted cross-entropy whose backward was already written over the logits by the forward kernel."""

    @staticmethod
    def forward(ctx, logits, labels, n_cols):
        loss, _ = ce_loss(logits, labels, n_cols, 1.0, want_grad=True)
        ctx.save_for_backward(logits)                   # now holds d(loss)/d(logits)
        return loss

    @staticmethod
    def backward(ctx, g):
        (dl,) = ctx.saved_tensors
        return dl * g.to(dl.dtype), None, None


class FusedHeadCEFn(torch.autograd.Function):
    """lm_head (DecoupledLinear, src/helpers.py:594-603) + HF shifted cross-entropy (train/losses.py:83-116) WITHOUT the
    [B, L, V] logits tensor: the rows are processed in chunks of `chunk` tokens - logits chunk (one two-segment GEMM straight
    off `weight[:n0]` and `additional_fc.weight`, no concatenated copy), loss rows; in the backward the chunk is recomputed,
    turned into d(logits) in place and consumed at once by the dgrad GEMM (d h) and the wgrad GEMMs, which accumulate into
    the two weights' gradient buffers.  Peak extra memory: one chunk (chunk x pad64(V) bf16) instead of B*L x V.
    Inputs: h [B, L, K] (after the final norm), weight [V0, K] (rows [:n0] used), add_w [n_add, K] or None, bias [V0] or
    None, add_b [n_add] or None, labels [B, L]."""

    @staticmethod
    def forward(ctx, h, weight, add_w, bias, add_b, labels, n0, chunk):
        _need_bf16(h, weight, add_w)
        B, Lq, K = h.shape
        n_add = 0 if add_w is None else add_w.shape[0]
        V = n0 + n_add
        Vp = _pad64(V)
        h2 = _rows2d(h)
        M = h2.shape[0]
        tgt = torch.full((B, Lq), -100, dtype=torch.int64, device=h.device)
        tgt[:, :-1] = labels[:, 1:].to(torch.int64)
        tgt = tgt.reshape(-1)
        nv = ((tgt >= 0) & (tgt < V)).sum().to(torch.int32).reshape(1)
        fb = None
        if bias is not None:                          # fused bias vector (64 KB - not the 197 MB weight)
            fb = torch.zeros((Vp,), dtype=h.dtype, device=h.device)
            fb[:n0] = bias.detach()[:n0]
            if add_b is not None:
                fb[n0:V] = add_b.detach()
        total = torch.zeros((), dtype=torch.float32, device=h.device)
        buf = torch.empty((min(chunk, M), Vp), dtype=h.dtype, device=h.device)
        for r0 in range(0, M, chunk):
            r1 = min(M, r0 + chunk)
            lg = buf[: r1 - r0]
            FusedHeadCEFn._logits(h2[r0:r1], weight, add_w, fb, n0, Vp, lg)
            total = total + ce_rows(lg, tgt[r0:r1], nv, V, want_grad=False).sum()
        ctx.save_for_backward(h, tgt, nv)
        ctx.fb, ctx.n0, ctx.chunk = fb, n0, chunk
        ctx.refs = (weight, add_w, bias, add_b)
        return total / nv.clamp(min=1).to(torch.float32)[0]

    @staticmethod
    def _logits(hc, weight, add_w, fb, n0, Vp, out):
        if add_w is None:           # plain nn.Linear head: columns [n0, Vp) of the chunk buffer stay unwritten (never read below V)
            return ops.linear(hc, weight, bias=fb, out=out[:, :n0])
        return ops.linear(hc, weight, bias=fb, out=out, w2=add_w, w2_row0=n0, n_rows=Vp)

    @staticmethod
    def backward(ctx, g):
        h, tgt, nv = ctx.saved_tensors
        w_ref, a_ref, b_ref, ab_ref = ctx.refs
        weight, add_w = w_ref, a_ref
        n0, chunk = ctx.n0, ctx.chunk
        n_add = 0 if add_w is None else add_w.shape[0]
        V = n0 + n_add
        Vp = _pad64(V)
        h2 = _rows2d(h)
        M, K = h2.shape
        need_h, need_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        need_a = add_w is not None and ctx.needs_input_grad[2]
        need_b = b_ref is not None and ctx.needs_input_grad[3]
        need_ab = ab_ref is not None and ctx.needs_input_grad[4]
        # transposed two-segment weight for the dgrad GEMM: [K, Vp], padding columns zero
        wt = None
        if need_h:
            wt = torch.zeros((K, Vp), dtype=h.dtype, device=h.device)
            transpose(weight[:n0], out=wt)                          # columns [0, pad64(n0)): rows n0.. of that range are zero-filled
            if n_add:
                wt[:, n0:V] = add_w.detach().t()
                if _pad64(n0) > V:
                    wt[:, V:_pad64(n0)] = 0
        # gradient destinations: the trainer's flat-buffer views when there are any (written in place, first chunk overwrites)
        def dest(param, rows):
            tgt_ = _fresh_target(param)
            live = param is not None and getattr(param, "_aki_grad_live", False)
            if tgt_ is None and not live:
                return torch.zeros_like(param), False, False          # plain autograd: a fresh tensor goes back
            return (param._aki_grad, True, live)
        gw = ga = gb = gab = None
        if need_w:
            gw, w_flat, w_live = dest(w_ref, n0)
            if not w_live and weight.shape[0] > n0:
                gw[n0:].zero_()                                      # rows of the original table the head never reads
        if need_a:
            ga, a_flat, a_live = dest(a_ref, n_add)
        dh = torch.empty_like(h2) if need_h else None
        dbias = torch.zeros((Vp,), dtype=torch.float32, device=h.device) if (need_b or need_ab) else None
        gs = g.to(h.dtype)
        buf = torch.empty((min(chunk, M), Vp), dtype=h.dtype, device=h.device)
        first = True
        for r0 in range(0, M, chunk):
            r1 = min(M, r0 + chunk)
            lg = buf[: r1 - r0]
            FusedHeadCEFn._logits(h2[r0:r1], weight, add_w, ctx.fb, n0, Vp, lg)
            ce_rows(lg, tgt[r0:r1], nv, V, want_grad=True)            # lg is now d(mean loss)/d(logits) on columns < V
            if Vp > V:
                lg[:, V:].zero_()
            lg.mul_(gs)
            if need_h:
                ops.linear(lg, wt, out=dh[r0:r1])
            if need_w or need_a:
                dT, xT = transpose(lg), transpose(h2[r0:r1])          # [Vp, pad64(rows)], [K, pad64(rows)]
                if need_w:
                    acc = (not first) or w_live
                    ops.linear(dT[:n0], xT, out=gw[:n0], residual=gw[:n0] if acc else None)
                if need_a:
                    acc = (not first) or a_live
                    ops.linear(dT[n0:V], xT, out=ga, residual=ga if acc else None)
            if dbias is not None:
                dbias += colsum(lg).float()
            first = False
        def finish(param, grad, flat):
            if not flat:
                return grad
            param._aki_grad_live = True
            hook = getattr(param, "_aki_grad_hook", None)
            if hook is not None:
                hook(param)
            return None
        out_w = finish(w_ref, gw, w_flat) if need_w else None
        out_a = finish(a_ref, ga, a_flat) if need_a else None
        out_b = out_ab = None
        if need_b:
            full = torch.zeros_like(b_ref)
            full[:n0] = dbias[:n0].to(full.dtype)
            out_b = _deliver(b_ref, lambda out: full if out is None else out.copy_(full))
        if need_ab:
            part = dbias[n0:V].to(ab_ref.dtype)
            out_ab = _deliver(ab_ref, lambda out: part if out is None else out.copy_(part))
        return (dh.view(h.shape) if need_h else None), out_w, out_a, out_b, out_ab, None, None, None


def fused_head_ce(h, head, labels, chunk: int = 2688):
    """Loss of `head` (nn.Linear or DecoupledLinear) on the final hidden states h [B, L, K] against labels [B, L], chunked."""
    if type(head) is torch.nn.Linear:
        return FusedHeadCEFn.apply(h, head.weight, None, head.bias, None, labels, head.weight.shape[0], chunk)
    n0 = head.max_original_id + 1
    add = head.additional_fc if head.additional_out_features else None
    bias = head.bias if (head.has_bias and head.bias is not None) else None
    return FusedHeadCEFn.apply(h, head.weight, None if add is None else add.weight, bias,
                               None if (add is None or bias is None) else add.bias, labels, n0, chunk)


class SpliceGradFn(torch.autograd.Function):
    """Backward of the language-stream splice (src/vlm.py:445-603 = torch.cat of embedding slices and vision tokens in the
    reference).  The forward output was produced by the HIP splice kernel; this node only routes d(inputs_embeds) to the
    vision tokens and to the rows of the two embedding tables (gather / index_add bookkeeping on precomputed indices)."""

    @staticmethod
    def forward(ctx, embeds, vision_tokens, embed_weight, embed_additional, lang_x, pos_lang, pos_vis, max_original_id):
        ctx.save_for_backward(lang_x, pos_lang, pos_vis)
        ctx.vshape, ctx.max_id = vision_tokens.shape, max_original_id
        ctx.w_ref, ctx.a_ref = embed_weight, embed_additional
        return embeds.view_as(embeds)

    @staticmethod
    def backward(ctx, g):
        lang_x, pos_lang, pos_vis = ctx.saved_tensors
        gf = g.reshape(-1, g.shape[-1])
        d_vis = d_w = d_a = None
        if ctx.needs_input_grad[1]:
            sel = gf.index_select(0, pos_vis.clamp(min=0).reshape(-1))
            d_vis = torch.where((pos_vis >= 0).reshape(-1, 1), sel, torch.zeros_like(sel)).view(ctx.vshape)
        if ctx.needs_input_grad[2] or (ctx.a_ref is not None and ctx.needs_input_grad[3]):
            keep = (pos_lang >= 0).reshape(-1)
            ids = lang_x.reshape(-1)[keep]
            rows = gf.index_select(0, pos_lang.reshape(-1)[keep])
            hi = ids > ctx.max_id
            if ctx.needs_input_grad[2]:
                def w_grad(out):
                    out = torch.zeros_like(ctx.w_ref) if out is None else out.zero_()
                    return _add_rows_deterministic(out, ids[~hi], rows[~hi])
                d_w = _deliver(ctx.w_ref, w_grad)
            if ctx.a_ref is not None and ctx.needs_input_grad[3]:
                def a_grad(out):
                    out = torch.zeros_like(ctx.a_ref) if out is None else out.zero_()
                    return _add_rows_deterministic(out, ids[hi] - ctx.max_id - 1, rows[hi])
                d_a = _deliver(ctx.a_ref, a_grad)
        return None, d_vis, d_w, d_a, None, None, None, None


def _add_rows_deterministic(out: torch.Tensor, idx: torch.Tensor, rows: torch.Tensor) -> torch.Tensor:
    """out[idx[i]] += rows[i] for a ZEROED `out`, reproducibly: index_add_ resolves repeated token ids with atomics, whose
    order (and therefore the bf16 rounding of every partial sum) changes from launch to launch - the only run-to-run
    difference tools/determinism_screen.py found in a training step.  Rows are sorted by id (stable), summed per id in f32
    in that order, rounded once and written to distinct rows."""
    if idx.numel() == 0:
        return out
    order = torch.argsort(idx, stable=True)
    uniq, counts = torch.unique_consecutive(idx[order], return_counts=True)
    sums = torch.segment_reduce(rows[order].float(), "sum", lengths=counts, axis=0)
    return out.index_copy_(0, uniq, sums.to(out.dtype))


def splice_positions(plan_h, lang_x_shape, n_img_max: int, Nv: int, L_out: int, padding_side: str, device):
    """From the host copy of the splice plan: where each original token and each vision vector landed in the output.
    pos_lang [B,T] / pos_vis [B,T_img,Nv]: flat row index into [B*L_out], -1 = not present."""
    import numpy as np
    B, Tn = lang_x_shape
    plan = plan_h.numpy()
    pos_lang = np.full((B, Tn), -1, dtype=np.int64)
    pos_vis = np.full((B, n_img_max, Nv), -1, dtype=np.int64)
    t = np.arange(Tn)
    for b in range(B):
        n_img, L_b = int(plan[b, 0]), int(plan[b, 2])
        tk = plan[b, 4:4 + n_img].astype(np.int64)
        off = (L_out - L_b) if padding_side == "left" else 0
        before = (t[:, None] > tk[None, :]).sum(1) if n_img else np.zeros(Tn, dtype=np.int64)
        p = off + t + before * (Nv - 1)
        is_img = np.isin(t, tk)
        pos_lang[b] = np.where(is_img, -1, b * L_out + p)
        for k in range(n_img):
            pos_vis[b, k] = b * L_out + off + tk[k] + k * (Nv - 1) + np.arange(Nv)
    return torch.from_numpy(pos_lang).to(device), torch.from_numpy(pos_vis).to(device)
