"""The pre-training step of BASELINE configs[2]: forward, backward, gradient all-reduce, clip
to 1.0 and AdamW - the body of ``train_one_epoch`` (train/train_utils.py:242-266) for a data-parallel replica
(train/train.py:311-312, 330-337), with every tensor op on HIP kernels.

Mixed precision the way bf16 autocast does it, made explicit: the optimizer owns fp32 master weights and moments in
flat buffers; the model's parameters are bf16 views into one flat bf16 buffer that the AdamW kernel rewrites after every
update; gradients are bf16 views into one flat buffer that the wgrad GEMMs write directly and RCCL all-reduces in place.
Per GPU for AKI-4B (3.90 B trainable parameters): 15.6 GB master + 31.2 GB moments + 7.8 GB weights + 7.8 GB gradients
(+7.8 GB transposed weights cached for the dgrad GEMMs) of the 288 GB HBM.
"""
from __future__ import annotations

from typing import Iterable, Optional

import torch
import torch.distributed as dist

from . import train_ops as T
from .dp import FlatGradReducer


class AkiTrainer:
    def __init__(self, model, lr: float = 1e-4, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.01,
                 max_grad_norm: float = 1.0, bucket_bytes: int = 512 << 20, group=None, shard_optimizer: bool = False,
                 exchange_when_alone: bool = False):
        """shard_optimizer: keep fp32 master weights and moments only for this rank's 1/world slice of every gradient
        bucket (reduce-scatter + all-gather instead of all-reduce) - the memory behaviour of the reference's FSDP launch
        configs (train/distributed.py:170-243, scripts/run_train.sh:23) for the optimizer state.
        exchange_when_alone: run the collectives even when the process group has a single rank (identities) - a one-GPU
        box then exercises the real RCCL reduce-scatter / all-gather / all-reduce entry points and stream hand-off."""
        self.model = model
        self.lr, self.betas, self.eps, self.weight_decay, self.max_grad_norm = lr, betas, eps, weight_decay, max_grad_norm
        self.step_count = 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        wd, nwd = model.group_params_by_weight_decay() if hasattr(model, "group_params_by_weight_decay") else (
            [p for p in model.parameters() if p.requires_grad], [])
        groups = [(list(wd), weight_decay), (list(nwd), 0.0)]
        dev = next(model.parameters()).device
        alone = bool(exchange_when_alone) and dist.is_initialized()
        self.shard = bool(shard_optimizer) and (self.world > 1 or alone)
        self.group = group
        # flat layout: [decay params | no-decay params], every parameter 16-byte aligned; when sharding, bucket ends (and
        # therefore segment ends) are padded to multiples of 8*world so every bucket splits evenly over the ranks
        align = 8 * self.world if self.shard else 8
        per = max(1, bucket_bytes // 2)
        spans, off, breaks = [], 0, []
        self.segments = []                       # (start, stop, weight_decay)
        for ps, decay in groups:
            start, bstart = off, off
            for p in ps:
                spans.append((p, off, off + p.numel()))
                off = (off + p.numel() + 7) // 8 * 8          # 16-byte aligned views
                if off - bstart >= per:                        # the reducer closes a bucket here: pad so it splits evenly
                    off = (off + align - 1) // align * align
                    bstart = off
            off = (off + align - 1) // align * align
            if off > start:
                self.segments.append((start, off, decay))
                breaks.append(off)
        n = off
        self.numel = n
        self.w16 = torch.zeros(n, dtype=torch.bfloat16, device=dev)
        self.g16 = torch.zeros(n, dtype=torch.bfloat16, device=dev)
        self.sqnorm = torch.zeros(1, dtype=torch.float32, device=dev)
        self.reducer = FlatGradReducer(self.g16, spans, bucket_bytes, group, shard=self.shard, breaks=breaks,
                                       exchange_when_alone=alone)
        self.params = []
        self.span_of = {id(p): (lo, hi) for p, lo, hi in spans}
        for p, lo, hi in spans:
            self.w16[lo:hi].copy_(p.detach().reshape(-1))
        # optimizer state: the slices this rank owns (everything when not sharding), stored compactly
        self.owned = []                          # (flat start, flat stop, state offset, weight_decay)
        so = 0
        for b in self.reducer.buckets:
            lo, hi = self.reducer.owned(b)
            decay = next(d for s0, s1, d in self.segments if s0 <= b[0] < s1)
            self.owned.append((lo, hi, so, decay))
            so += hi - lo
        self.master = torch.zeros(so, dtype=torch.float32, device=dev)
        self.m = torch.zeros(so, dtype=torch.float32, device=dev)
        self.v = torch.zeros(so, dtype=torch.float32, device=dev)
        for p, lo, hi in spans:                  # fp32 master copy of what this rank owns (from the incoming weights)
            src = p.detach().reshape(-1).float()
            for olo, ohi, so_, _ in self.owned:
                a_, b_ = max(lo, olo), min(hi, ohi)
                if a_ < b_:
                    self.master[so_ + a_ - olo: so_ + b_ - olo].copy_(src[a_ - lo: b_ - lo])
        for p, lo, hi in spans:
            p.data = self.w16[lo:hi].view(p.shape)
            p._aki_grad = self.g16[lo:hi].view(p.shape)
            p._aki_grad_live = False
            p._aki_grad_hook = self.reducer.notify
            p.grad = None
            self.params.append(p)
        # everything that is not trained (the frozen vision tower) is read as bf16 as well
        for p in model.parameters():
            if not p.requires_grad and p.dtype != torch.bfloat16:
                p.data = p.data.to(torch.bfloat16)
        for b_ in model.buffers():
            if b_.dtype == torch.float32 and b_.dim() > 0:
                pass                               # RoPE tables etc. stay f32 (the kernels take f32 cos/sin)
        T.bump_weight_epoch()

    # ---- one step ---------------------------------------------------------------------------------------------------
    def zero_grad(self) -> None:
        for p in self.params:
            p._aki_grad_live = False
            p.grad = None

    def backward(self, loss: torch.Tensor, last_microbatch: bool = True) -> None:
        """loss.backward() through the HIP kernels.  Gradient accumulation (train/train_utils.py:242-266 divides the loss
        by `gradient_accumulation_steps` and steps every k-th micro-batch): call zero_grad() once, then backward(loss / k,
        last_microbatch=False) for the first k-1 micro-batches - gradients add up in the flat buffer, nothing is
        exchanged - and backward(loss / k) for the last one, which also runs the (overlapped) gradient exchange."""
        self.reducer.enabled = bool(last_microbatch)
        loss.backward()
        for p in self.params:
            if p.grad is not None:               # a gradient autograd produced itself (no HIP writer took it): fold it in
                if p._aki_grad_live:
                    p._aki_grad += p.grad.to(torch.bfloat16)
                else:
                    p._aki_grad.copy_(p.grad)
                    p._aki_grad_live = True
                p.grad = None
                self.reducer.notify(p)
            elif not p._aki_grad_live:           # unused so far in this window
                if last_microbatch:
                    p._aki_grad.zero_()
                    self.reducer.notify(p)
        if last_microbatch:
            self.reducer.finish()
        self.reducer.enabled = True

    def optimizer_step(self) -> None:
        self.step_count += 1
        gscale = 1.0 / self.world
        first = True
        for lo, hi, _, _ in self.owned:
            T.grad_sqnorm(self.g16[lo:hi], self.sqnorm, accumulate=not first)
            first = False
        if self.shard:                           # every rank holds the sum over its own slices: one scalar all-reduce
            dist.all_reduce(self.sqnorm, op=dist.ReduceOp.SUM, group=self.group)
        for lo, hi, so, decay in self.owned:
            n = hi - lo
            T.adamw_step(self.master[so:so + n], self.m[so:so + n], self.v[so:so + n], self.g16[lo:hi], self.w16[lo:hi],
                         self.sqnorm, self.max_grad_norm, gscale, self.lr, self.betas[0], self.betas[1], self.eps, decay,
                         self.step_count)
        self.reducer.all_gather_weights(self.w16)
        T.bump_weight_epoch()

    # ---- checkpoint / resume (train/train_utils.py:395-460 saves model + optimizer state; the I/O itself is out of scope) ----
    def refresh_master(self) -> None:
        """Re-derive the fp32 master weights from the model's current bf16 weights (after model.load_state_dict)."""
        for lo, hi, so, _ in self.owned:
            self.master[so: so + hi - lo].copy_(self.w16[lo:hi])
        T.bump_weight_epoch()

    def state_dict(self) -> dict:
        """Optimizer state of THIS rank (all of it when the optimizer is not sharded): fp32 master weights, both moments and
        the step counter, as flat tensors in the trainer's parameter order."""
        return {"step": self.step_count, "numel": self.numel, "sharded": self.shard, "owned": [(lo, hi) for lo, hi, _, _ in self.owned],
                "master": self.master.clone(), "exp_avg": self.m.clone(), "exp_avg_sq": self.v.clone()}

    def load_state_dict(self, sd: dict) -> None:
        if sd["numel"] != self.numel or [tuple(o) for o in sd["owned"]] != [(lo, hi) for lo, hi, _, _ in self.owned]:
            raise ValueError("optimizer state does not match this model / sharding layout")
        self.step_count = int(sd["step"])
        self.master.copy_(sd["master"])
        self.m.copy_(sd["exp_avg"])
        self.v.copy_(sd["exp_avg_sq"])
        for lo, hi, so, _ in self.owned:                    # the forward reads the bf16 image of the restored master weights
            self.w16[lo:hi].copy_(self.master[so: so + hi - lo])
        self.reducer.all_gather_weights(self.w16)
        T.bump_weight_epoch()

    def grad_norm(self) -> torch.Tensor:
        """Global L2 norm of the (averaged) gradients of the last step, as clip_grad_norm_ returns it."""
        return self.sqnorm.sqrt() / self.world

    def train_step(self, vision_x, lang_x, attention_mask=None, labels=None) -> torch.Tensor:
        """forward -> backward (+ overlapped gradient all-reduce) -> clip -> AdamW.  Returns the detached loss."""
        self.zero_grad()
        out = self.model(vision_x, lang_x, attention_mask=attention_mask, labels=labels)
        loss = out[0] if not hasattr(out, "loss") else out.loss
        self.backward(loss)
        self.optimizer_step()
        return loss.detach()
