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
    _need_bf16(g)
    lib = L.load()
    ws = _ws(lib.aki_grad_sqnorm_workspace_bytes(), _dev(g, out))
    L.check(lib.aki_grad_sqnorm(_ptr(g), g.numel(), _ptr(out), 1 if accumulate else 0, _BF16, _ptr(ws), ws.numel(), _stream()),
            "aki_grad_sqnorm")


def adamw_step(p32, m, v, g16, w16, sqnorm, max_norm, gscale, lr, beta1, beta2, eps, wd, step) -> None:
    _dev(p32, m, v, g16, w16, sqnorm)
    L.check(L.load().aki_adamw_step(_ptr(p32), _ptr(m), _ptr(v), _ptr(g16), _ptr(w16), p32.numel(), _ptr(sqnorm), float(max_norm),
                                    float(gscale), float(lr), float(beta1), float(beta2), float(eps), float(wd), int(step), _stream()),
            "aki_adamw_step")


# ---- transposed-weight cache -------------------------------------------------------------------------------------
_EPOCH = 0            # bumped by the trainer after every optimizer step (the kernels write weights through raw pointers)
_WT = {}


def bump_weight_epoch() -> None:
    global _EPOCH
    _EPOCH += 1
    _WT.clear()


def _weight_t(w: torch.Tensor) -> torch.Tensor:
    """W [N,K] -> W^T [K, pad64(N)] (zero padded), cached until the weights change: the trainer bumps the epoch after every
    optimizer step (its kernels write through raw pointers), in-place torch updates show up in `_version`.  Temporaries (the
    lm_head's concatenated weight is a new tensor every forward) would pile up without a trainer, hence the size cap."""
    if not (isinstance(w, torch.nn.Parameter) or hasattr(w, "_aki_grad") or getattr(w, "_aki_cacheable", False)):
        # a temporary (slice / concatenation built inside a forward): its (data_ptr, _version) says nothing about its content -
        # the allocator recycles the address and a fresh tensor is always version 0 - so it is transposed every time
        return transpose(w if w.stride(1) == 1 else w.contiguous())
    key = (w.data_ptr(), tuple(w.shape), w._version, _EPOCH)
    t = _WT.get(key)
    if t is None:
        if len(_WT) >= 320:                  # > every 2-D weight of AKI-4B (32 x 4 + connector + heads)
            _WT.clear()
        t = transpose(w if w.stride(1) == 1 else w.contiguous())
        _WT[key] = t
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


# ---- autograd Functions ---------------------------------------------------------------------------------------------
class LinearFn(torch.autograd.Function):
    """y = x W^T + b [+ residual]   (HIP MFMA GEMM both ways: dX = dY W, dW = dY^T X, db = colsum dY)."""

    @staticmethod
    def forward(ctx, x, w, bias, residual):
        _need_bf16(x, w)
        y = ops.linear(x, w, bias=bias, residual=residual)
        ctx.save_for_backward(x, w)
        ctx.has_bias, ctx.has_res = bias is not None, residual is not None
        ctx.bias_ref = bias
        ctx.w_ref = w
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
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
            dyT, xT = transpose(dy2), transpose(x2)       # [N, Mp], [K, Mp]
            dw = _deliver(ctx.w_ref, lambda out: ops.linear(dyT, xT, out=out))
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = _deliver(ctx.bias_ref, lambda out: colsum(dy2) if out is None else out.copy_(colsum(dy2)))
        return dx, dw, db, (dy if ctx.has_res else None)


def linear(x, w, bias=None, residual=None):
    return LinearFn.apply(x, w, bias, residual)


class NormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, eps, rms):
        y = ops.rmsnorm(x, w, eps) if rms else ops.layernorm(x, w, b, eps)
        ctx.save_for_backward(x, w)
        ctx.eps, ctx.rms, ctx.w_ref, ctx.b_ref = eps, rms, w, b
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
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
        ctx.save_for_backward(x, w)
        ctx.eps, ctx.w_ref = eps, w
        return y, x.view_as(x)

    @staticmethod
    def backward(ctx, dy, dres):
        x, w = ctx.saved_tensors
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
        ctx.save_for_backward(x, w, cos, sin)
        ctx.pos, ctx.w_ref = position_ids, w
        return q, k, v

    @staticmethod
    def backward(ctx, dq, dk, dv):
        x, w, cos, sin = ctx.saved_tensors
        dqkv = rope_bwd_merge(dq, dk, dv, cos, sin, ctx.pos)          # [B, L, 3*H*Dh]
        d2 = dqkv.view(-1, dqkv.shape[-1])
        dx = ops.linear(d2, _weight_t(w)).view(x.shape) if ctx.needs_input_grad[0] else None
        dw = None
        if ctx.needs_input_grad[1]:
            dT, xT = transpose(d2), transpose(_rows2d(x))
            dw = _deliver(ctx.w_ref, lambda out: ops.linear(dT, xT, out=out))
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
