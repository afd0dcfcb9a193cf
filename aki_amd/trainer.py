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
                 max_grad_norm: float = 1.0, bucket_bytes: int = 512 << 20, group=None, shard_optimizer: Optional[bool] = None,
                 exchange_when_alone: bool = False, clip_every_microbatch: bool = False, reduce_dtype: Optional[torch.dtype] = None,
                 emit_transposes: bool = True, first_bucket_bytes: Optional[int] = 64 << 20):
        """first_bucket_bytes: the first gradient bucket of each segment - the parameters whose gradients the backward pass finishes LAST - is
        this small, so that the exchange left over after the backward pass is short (dp.FlatGradReducer); None = bucket_bytes.
        shard_optimizer: keep fp32 master weights and moments only for this rank's 1/world slice of every gradient
        bucket (reduce-scatter + all-gather instead of all-reduce) - the memory behaviour of the reference's FSDP launch
        configs (train/distributed.py:170-243, scripts/run_train.sh:23) for the optimizer state.  Default (None): ON whenever
        there is more than one rank - the AdamW pass is HBM-bound (21 ms of a 150 ms step at world 1, 24 bytes per parameter) and
        sharding is the only thing that shrinks it (projected 2.6 ms at 8 ranks) at the same exchanged bytes; False restores the
        reference's plain DDP replica (every rank updates everything).
        reduce_dtype: None / bfloat16 exchanges the bf16 gradients where the backward kernels left them (the reference's FSDP
        default, train/distributed.py:160-167 `reduce_dtype=bf16`); float32 widens every bucket into an fp32 buffer, sums in fp32
        across ranks (and across the micro-batches of an accumulation window) and lets the optimizer consume the fp32 sum, at twice
        the exchanged bytes and one more flat buffer (4 bytes per parameter).  This is an fp32 SUM OF bf16 LOCAL GRADIENTS: each
        rank's backward kernels still write their gradient - including the in-backward accumulation of a weight used twice - rounded
        to bf16.  The reference's DDP path under `--precision amp_bf16` (train/train.py:311-312, train/train_utils.py:56-65) keeps
        fp32 local .grad as well; that part is not reproduced.
        With a sharded optimizer (the default for world > 1) `state_dict()` is THIS RANK's slice: a checkpoint written by rank 0 alone
        would drop (world - 1) / world of the optimizer state - save `full_state_dict()` (a collective that gathers master weights and
        moments) instead, and restore it with `load_full_state_dict()` under any sharding layout; `emit_transposes` has no effect under
        sharding (a rank does not update whole weights).
        exchange_when_alone: run the collectives even when the process group has a single rank (identities) - a one-GPU
        box then exercises the real RCCL reduce-scatter / all-gather / all-reduce entry points and stream hand-off.
        clip_every_microbatch: single-rank parity only (see backward()).
        emit_transposes: (unsharded optimizer only) the AdamW pass of every 2-D weight also writes W^T, the operand of the backward's
        input-gradient GEMMs, into a buffer this trainer owns - instead of one aki_transpose launch per weight after every optimizer step
        (a read and a write of all 7.8 GB of bf16 weights: 3 ms of the 150 ms step).  Same arithmetic: weights bit-identical either way."""
        self.model = model
        if getattr(model, "_gradient_checkpointing", False) and hasattr(model, "init_gradient_checkpointing"):
            model.init_gradient_checkpointing()  # what the reference's driver does after wrapping the model (train/train.py:315-327)
        self.lr, self.betas, self.eps, self.weight_decay, self.max_grad_norm = lr, betas, eps, weight_decay, max_grad_norm
        self.step_count = 0
        self.gacc = None                         # fp32 gradient accumulator (gradient accumulation windows only)
        self._acc_open = False
        self.clip_every_microbatch = bool(clip_every_microbatch)
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        if shard_optimizer is None:
            shard_optimizer = self.world > 1
        if reduce_dtype not in (None, torch.bfloat16, torch.float32):
            raise ValueError("reduce_dtype must be None, torch.bfloat16 or torch.float32")
        self.reduce_dtype = torch.float32 if reduce_dtype == torch.float32 else torch.bfloat16
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
        per_first = per if first_bucket_bytes is None else max(1, min(per, first_bucket_bytes // 2))
        spans, off, breaks = [], 0, []
        self.segments = []                       # (start, stop, weight_decay)
        for ps, decay in groups:
            start, bstart = off, off
            for p in ps:
                spans.append((p, off, off + p.numel()))
                off = (off + p.numel() + 7) // 8 * 8          # 16-byte aligned views
                if off - bstart >= (per_first if bstart == start else per):     # the reducer closes a bucket here: pad so it splits evenly
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
        # fp32 exchange: g32 receives each bucket of g16 widened, is summed across ranks in fp32 and feeds the optimizer; it doubles
        # as the fp32 accumulator of gradient-accumulation windows
        self.g32 = torch.zeros(n, dtype=torch.float32, device=dev) if self.reduce_dtype == torch.float32 else None
        self.reducer = FlatGradReducer(self.g16, spans, bucket_bytes, group, shard=self.shard, breaks=breaks,
                                       exchange_when_alone=alone, flat32=self.g32, first_bucket_bytes=first_bucket_bytes)
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
        # W^T emitted by the optimizer pass: 2-D trainable weights with K % 4 == 0, when this rank updates every element itself
        self.t_jobs = []                         # (param, flat lo, flat hi, N, K, W^T view)
        if emit_transposes and not self.shard:
            tot = 0
            elig = []
            # the weights the backward transposes: nn.Linear weights that go through train_ops.linear / QkvRopeFn (not embeddings; not the
            # lm_head, whose chunked head + loss node reads its weight segments in place)
            lin = {id(m.weight) for n_, m in model.named_modules() if isinstance(m, torch.nn.Linear) and "lm_head" not in n_}
            for p, lo, hi in spans:
                if id(p) in lin and p.dim() == 2 and p.shape[1] % 4 == 0 and lo % 8 == 0:
                    N_, K_ = int(p.shape[0]), int(p.shape[1])
                    ldT = (N_ + 63) // 64 * 64
                    elig.append((p, lo, hi, N_, K_, tot, ldT))
                    tot += K_ * ldT
            if elig:
                self.wT = torch.zeros(tot, dtype=torch.bfloat16, device=dev)
                self.t_jobs = [(p, lo, hi, N_, K_, self.wT[o: o + K_ * ldT].view(K_, ldT)) for p, lo, hi, N_, K_, o, ldT in elig]
        T.bump_weight_epoch()

    # ---- one step ---------------------------------------------------------------------------------------------------
    def zero_grad(self) -> None:
        self._acc_open = False
        for p in self.params:
            p._aki_grad_live = False
            p.grad = None

    def backward(self, loss: torch.Tensor, last_microbatch: bool = True) -> None:
        """loss.backward() through the HIP kernels.  Gradient accumulation (train/train_utils.py:242-266 divides the loss
        by `gradient_accumulation_steps` and steps every k-th micro-batch): call zero_grad() once, then backward(loss / k,
        last_microbatch=False) for the first k-1 micro-batches and backward(loss / k) for the last one.

        Like the reference (fp32 parameter .grad under autocast; the sharded fp32 gradient of its FSDP wrap,
        train/distributed.py:160-167) the micro-batch gradients are SUMMED IN FP32: every micro-batch writes its own bf16
        gradients into the flat buffer (first writer overwrites), which is then added to an fp32 accumulator of the same
        length (allocated on the first accumulation window; 15.6 GB for AKI-4B).  The last micro-batch rounds the fp32 sum to
        bf16 ONCE, into the flat buffer, and exchanges that (after the backward pass - the exchange is linear, but overlapping
        it with this backward would exchange the last micro-batch alone).  `clip_every_microbatch` additionally clips the
        accumulated fp32 gradient to `max_grad_norm` after every micro-batch, which is where the reference's loop calls
        clip_grad_norm_ (train/train_utils.py:254-258: after each backward, not once per optimizer step); the default
        clips once, in the optimizer step.  SINGLE-RANK parity only: the reference clips gradients its DDP / FSDP wrap has
        already reduced across ranks, this flag clips the rank-LOCAL accumulator before any exchange - with more than one rank
        the norms and clip factors differ per rank and the clip precedes the average.  With gradient_accumulation_steps = 1
        nothing of this runs."""
        accumulating = (not last_microbatch) or self._acc_open
        self.reducer.enabled = bool(last_microbatch) and not accumulating
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
            elif not p._aki_grad_live:           # unused in this backward pass
                if last_microbatch or accumulating:
                    p._aki_grad.zero_()
                    self.reducer.notify(p)
        if accumulating:
            if self.g32 is not None:
                self.gacc = self.g32             # fp32 exchange: the accumulator IS the exchanged buffer
            elif self.gacc is None:
                self.gacc = torch.empty(self.numel, dtype=torch.float32, device=self.g16.device)
            if self._acc_open:
                self.gacc.add_(self.g16)         # fp32 += bf16 (exact conversion, fp32 sum)
            else:
                self.gacc.copy_(self.g16)
                self._acc_open = True
            if self.clip_every_microbatch and self.max_grad_norm is not None and self.max_grad_norm > 0:
                norm = self.gacc.norm()
                self.gacc.mul_(torch.clamp(self.max_grad_norm / (norm + 1e-6), max=1.0))     # torch.nn.utils.clip_grad_norm_
            if last_microbatch:
                if self.g32 is None:
                    self.g16.copy_(self.gacc)    # ONE rounding of the fp32 sum
                self._acc_open = False
                self.reducer.enabled = True
                self.reducer.copy_in = False     # (fp32 exchange) g32 already holds the window's sum
                for p in self.params:
                    p._aki_grad_live = True
                    self.reducer.notify(p)
            else:
                for p in self.params:            # the next micro-batch's first writers overwrite their slices again
                    p._aki_grad_live = False
        if last_microbatch:
            self.reducer.finish()
            self.reducer.copy_in = True
        self.reducer.enabled = True

    def optimizer_step(self) -> None:
        self.step_count += 1
        gscale = 1.0 / self.world
        g = self.g16 if self.g32 is None else self.g32      # fp32 exchange: the optimizer consumes the fp32 sum
        first = True
        for lo, hi, _, _ in self.owned:
            T.grad_sqnorm(g[lo:hi], self.sqnorm, accumulate=not first)
            first = False
        if self.shard:                           # every rank holds the sum over its own slices: one scalar all-reduce
            dist.all_reduce(self.sqnorm, op=dist.ReduceOp.SUM, group=self.group)
        hp = (self.sqnorm, self.max_grad_norm, gscale, self.lr, self.betas[0], self.betas[1], self.eps)
        if self.t_jobs:
            # unsharded: state offsets = flat offsets.  2-D weights go through the transposing kernel one by one, everything between them
            # (1-D parameters, alignment gaps) through the flat kernel, segment by segment (weight decay differs)
            jobs = iter(self.t_jobs)
            job = next(jobs, None)
            for s0, s1, decay in self.segments:
                at = s0
                while job is not None and job[1] < s1:
                    p_, lo, hi, N_, K_, wT = job
                    if lo > at:
                        T.adamw_step(self.master[at:lo], self.m[at:lo], self.v[at:lo], g[at:lo], self.w16[at:lo], *hp, decay, self.step_count)
                    T.adamw_step_t(self.master[lo:hi], self.m[lo:hi], self.v[lo:hi], g[lo:hi], self.w16[lo:hi], wT, N_, K_, *hp, decay,
                                   self.step_count)
                    at = (hi + 7) // 8 * 8
                    job = next(jobs, None)
                if s1 > at:
                    T.adamw_step(self.master[at:s1], self.m[at:s1], self.v[at:s1], g[at:s1], self.w16[at:s1], *hp, decay, self.step_count)
        else:
            for lo, hi, so, decay in self.owned:
                n = hi - lo
                T.adamw_step(self.master[so:so + n], self.m[so:so + n], self.v[so:so + n], g[lo:hi], self.w16[lo:hi], *hp, decay,
                             self.step_count)
        self.reducer.all_gather_weights(self.w16)
        if self.g32 is None:
            self.gacc = None                     # the fp32 accumulator of a finished window (15.6 GB for AKI-4B) is not kept
        T.bump_weight_epoch()
        for p_, lo, hi, N_, K_, wT in self.t_jobs:             # the transposes of the weights just written
            T.register_weight_t(p_, wT)

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

    def full_state_dict(self, cpu: bool = True) -> dict:
        """COLLECTIVE (every rank calls it): the whole optimizer state - fp32 master weights and both moments over the full flat layout -
        on every rank, whatever the sharding; what rank 0 should write into a checkpoint (the reference saves a consolidated optimizer
        state through FSDP.optim_state_dict, train/train_utils.py:292-327).  One flat tensor at a time: 4 bytes per parameter in flight."""
        out = {"step": self.step_count, "numel": self.numel, "full": True}
        for name, src in (("master", self.master), ("exp_avg", self.m), ("exp_avg_sq", self.v)):
            full = torch.zeros(self.numel, dtype=torch.float32, device=src.device)
            for lo, hi, so, _ in self.owned:
                full[lo:hi].copy_(src[so: so + hi - lo])
            if self.shard and self.world > 1:
                dist.all_reduce(full, op=dist.ReduceOp.SUM, group=self.group)      # every element is owned by exactly one rank
            out[name] = full.cpu() if cpu else full
        return out

    def load_full_state_dict(self, sd: dict) -> None:
        """Restore from `full_state_dict()` under THIS trainer's sharding layout (which may differ from the saving run's)."""
        if not sd.get("full") or sd["numel"] != self.numel:
            raise ValueError("not a consolidated optimizer state of this model (full_state_dict())")
        self.step_count = int(sd["step"])
        for name, dst in (("master", self.master), ("exp_avg", self.m), ("exp_avg_sq", self.v)):
            for lo, hi, so, _ in self.owned:
                dst[so: so + hi - lo].copy_(sd[name][lo:hi])
        for lo, hi, so, _ in self.owned:
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


# ======================================================================================================================
# Parameter-sharded data parallelism: the FSDP FULL_SHARD exchange of the reference's shipped launch
# configurations (train/distributed.py:170-243, wrap units from `get_fsdp_lambda_fn`, scripts/run_train.sh:23).
# ======================================================================================================================
def _storage_bytes(t: torch.Tensor) -> int:
    return t.untyped_storage().size()


class _PreBackward(torch.autograd.Function):
    """Identity placed on a unit's output: its backward runs before anything inside the unit does, which is where the unit's
    parameters are gathered again and its gradient buffer is allocated."""

    @staticmethod
    def forward(ctx, y, unit):
        ctx.unit = unit
        return y.view_as(y)

    @staticmethod
    def backward(ctx, dy):
        ctx.unit.materialize()
        ctx.unit.alloc_grads()
        return dy, None


class _Unit:
    """One sharding unit: the trainable parameters of a decoder block / the vision tokenizer / the root remainder as ONE flat
    bf16 buffer whose storage only exists while the unit computes.  Each rank keeps 1/world of it (bf16 shard, fp32 master
    weights and moments of that shard) - weights, gradients and optimizer state are all sharded."""

    def __init__(self, name, params, decay, world, rank, group, device, no_scatter):
        self.name, self.params, self.decay = name, params, decay
        self.world, self.rank, self.group, self.no_scatter = world, rank, group, no_scatter
        offs, off = [], 0
        for p in params:
            offs.append(off)
            off = (off + p.numel() + 7) // 8 * 8
        align = 8 * world
        self.numel = (off + align - 1) // align * align
        self.n_shard = self.numel // world
        self.lo = rank * self.n_shard
        self.w_full = torch.zeros(self.numel, dtype=torch.bfloat16, device=device)
        self.g_full = torch.zeros(self.numel, dtype=torch.bfloat16, device=device)
        self.nbytes = self.numel * 2
        for p, o in zip(params, offs):
            self.w_full[o:o + p.numel()].copy_(p.detach().reshape(-1))
        self.w_shard = self.w_full[self.lo:self.lo + self.n_shard].clone()
        self.g_shard = torch.zeros(self.n_shard, dtype=torch.bfloat16, device=device)
        self.master = self.w_shard.float()
        self.m = torch.zeros_like(self.master)
        self.v = torch.zeros_like(self.master)
        for p, o in zip(params, offs):
            p.data = self.w_full[o:o + p.numel()].view(p.shape)
            p._aki_grad = self.g_full[o:o + p.numel()].view(p.shape)
            p._aki_grad_live = False
            p._aki_unit = self
            p.grad = None
        self.pending = len(params)
        self.delivered = set()
        self.inflight = None               # (work, tmp shard) of the reduce-scatter in flight
        self.fresh = True                  # g_shard holds nothing of this accumulation window yet
        self.g_acc32 = None                # fp32 running sum of the window (second micro-batch onwards)
        self.release()
        self.release_grads()

    # -- weights -------------------------------------------------------------------------------------------------------
    def live(self) -> bool:
        return _storage_bytes(self.w_full) != 0

    def materialize(self) -> None:
        """All-gather the unit's bf16 weights from the ranks' shards (no-op while the storage is alive)."""
        if self.live():
            return
        self.w_full.untyped_storage().resize_(self.nbytes)
        mine = self.w_full[self.lo:self.lo + self.n_shard]
        if self.world == 1 and not dist.is_initialized():
            mine.copy_(self.w_shard)
        elif not self.no_scatter:
            dist.all_gather_into_tensor(self.w_full, self.w_shard, group=self.group)
        else:                              # gloo: zero the foreign slices and sum
            self.w_full.zero_()
            mine.copy_(self.w_shard)
            dist.all_reduce(self.w_full, op=dist.ReduceOp.SUM, group=self.group)

    def release(self) -> None:
        if self.live():
            self.w_full.untyped_storage().resize_(0)

    # -- gradients ---------------------------------------------------------------------------------------------------------
    def alloc_grads(self) -> None:
        if _storage_bytes(self.g_full) == 0:
            self.g_full.untyped_storage().resize_(self.nbytes)
            self.g_full.zero_()            # padding between parameters and parameters nobody writes must contribute zeros
            for p in self.params:
                p._aki_grad_live = False

    def release_grads(self) -> None:
        if _storage_bytes(self.g_full) != 0:
            self.g_full.untyped_storage().resize_(0)

    def reduce_grads(self) -> None:
        """Reduce-scatter the unit's gradients: rank r receives the SUM of slice r, added to its gradient shard."""
        tmp = torch.empty_like(self.g_shard)
        if self.world == 1 and not dist.is_initialized():
            tmp.copy_(self.g_full[self.lo:self.lo + self.n_shard])
            work = None
        elif not self.no_scatter:
            work = dist.reduce_scatter_tensor(tmp, self.g_full, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        else:
            work = dist.all_reduce(self.g_full, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self.inflight = (work, tmp)

    def finish_reduce(self) -> None:
        """Wait for the reduce-scatter in flight (the compute stream waits, not the host), fold it into the shard, free the
        unit's full gradient buffer and - when nothing else needs them - its weights."""
        if self.inflight is None:
            return
        work, tmp = self.inflight
        if work is not None:
            work.wait()
        if self.no_scatter and dist.is_initialized():
            tmp.copy_(self.g_full[self.lo:self.lo + self.n_shard])
        if self.fresh:
            self.g_shard.copy_(tmp)
            self.fresh = False
            self.g_acc32 = None
        else:
            # later micro-batch of an accumulation window: the running sum is kept in fp32 (the reference's FSDP wrap keeps its
            # sharded gradient in fp32, train/distributed.py:160-167) and g_shard is its bf16 image, rounded once from the sum
            if self.g_acc32 is None:
                self.g_acc32 = self.g_shard.float()
            self.g_acc32 += tmp
            self.g_shard.copy_(self.g_acc32)
        self.inflight = None
        self.release_grads()
        self.release()


class AkiShardedTrainer:
    """AkiTrainer's step (forward, backward, clip, AdamW; train/train_utils.py:242-266) with PARAMETERS sharded as well -
    what FSDP(FULL_SHARD) does for the reference (train/distributed.py:193-222): every rank stores 1/world of each unit's
    bf16 weights, gradients, fp32 master weights and moments; a unit's weights are all-gathered right before its forward,
    released after it, gathered again right before its backward, and its gradients are reduce-scattered as soon as its
    last wgrad GEMM has written them (overlapping the next unit's backward; the wait is deferred by one unit).
    Units = the modules `model.get_fsdp_lambda_fn()` selects (decoder blocks, vision tokenizer) + the remaining trainable
    parameters as root units (alive for the whole step, one per weight-decay group)."""

    def __init__(self, model, lr: float = 1e-4, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.01,
                 max_grad_norm: float = 1.0, group=None):
        self.model, self.group = model, group
        self.lr, self.betas, self.eps, self.weight_decay, self.max_grad_norm = lr, betas, eps, weight_decay, max_grad_norm
        self.step_count = 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        no_scatter = dist.is_initialized() and str(dist.get_backend(group)).lower() == "gloo"
        dev = next(model.parameters()).device
        decayed, plain = model.group_params_by_weight_decay()
        plain_ids = {id(p) for p in plain}
        is_unit = model.get_fsdp_lambda_fn()
        self.units, taken = [], set()
        self._hooks = []
        for name, mod in model.named_modules():
            if not is_unit(mod):
                continue
            ps = [p for p in mod.parameters() if p.requires_grad and id(p) not in taken]
            if not ps:
                continue
            assert not any(id(p) in plain_ids for p in ps), "a sharding unit with mixed weight decay"
            taken.update(id(p) for p in ps)
            u = _Unit(name, ps, weight_decay, self.world, self.rank, group, dev, no_scatter)
            self.units.append(u)
            self._hooks.append(mod.register_forward_pre_hook(lambda m, a, u=u: u.materialize()))
            self._hooks.append(mod.register_forward_hook(lambda m, a, out, u=u: self._after_forward(u, out)))
        self.roots = []
        for nm, ps, dec in (("root", [p for p in decayed if id(p) not in taken], weight_decay), ("root.embeddings", list(plain), 0.0)):
            ps = [p for p in ps if p.requires_grad]
            if ps:
                self.roots.append(_Unit(nm, ps, dec, self.world, self.rank, group, dev, no_scatter))
        self.all_units = self.units + self.roots
        self.params = [p for u in self.all_units for p in u.params]
        for p in self.params:
            p._aki_grad_hook = self.notify
        for p in model.parameters():
            if not p.requires_grad and p.dtype != torch.bfloat16:
                p.data = p.data.to(torch.bfloat16)
        self._hooks.append(model.register_forward_pre_hook(self._before_model_forward))
        self._hooks.append(model.register_forward_hook(self._after_model_forward))
        self.sqnorm = torch.zeros(1, dtype=torch.float32, device=dev)
        self.numel = sum(u.numel for u in self.all_units)
        self._last = None                  # unit whose reduce-scatter is in flight
        self.enabled = True
        T.bump_weight_epoch()            # (cached transforms of these weights: train_ops.sharded() keeps them out of the caches)

    # ---- hooks ---------------------------------------------------------------------------------------------------------
    def _after_forward(self, u, out):
        if torch.is_grad_enabled() and isinstance(out, torch.Tensor) and out.requires_grad:
            out = _PreBackward.apply(out, u)
        u.release()
        return out

    def _before_model_forward(self, module, args):
        for r in self.roots:
            r.materialize()
        return None

    def _after_model_forward(self, module, args, out):
        if not torch.is_grad_enabled():
            for r in self.roots:
                r.release()
        return None

    def notify(self, p) -> None:
        """A parameter's gradient slice is final for this backward pass."""
        u = p._aki_unit
        if id(p) in u.delivered:
            return
        u.delivered.add(id(p))
        u.pending -= 1
        if u.pending == 0 and self.enabled:
            self._launch(u)

    def _launch(self, u) -> None:
        if self._last is not None:
            self._last.finish_reduce()     # deferred by one unit: that exchange ran under this unit's backward
        u.reduce_grads()
        self._last = u

    # ---- one step ------------------------------------------------------------------------------------------------------
    def zero_grad(self) -> None:
        for u in self.all_units:
            u.fresh = True
            for p in u.params:
                p._aki_grad_live = False
                p.grad = None

    def backward(self, loss: torch.Tensor) -> None:
        for r in self.roots:
            r.materialize()
            r.alloc_grads()
        for u in self.all_units:
            u.pending, u.delivered = len(u.params), set()
        loss.backward()
        for u in self.all_units:
            if u.pending == 0:
                continue
            u.alloc_grads()                # a unit nothing flowed through (or whose gradients come from autograd itself)
            for p in u.params:
                if p.grad is not None:
                    if p._aki_grad_live:
                        p._aki_grad += p.grad.to(torch.bfloat16)
                    else:
                        p._aki_grad.copy_(p.grad)
                        p._aki_grad_live = True
                    p.grad = None
                self.notify(p)
        if self._last is not None:
            self._last.finish_reduce()
            self._last = None
        for u in self.all_units:
            u.release_grads()
            u.release()

    def optimizer_step(self) -> None:
        self.step_count += 1
        for i, u in enumerate(self.all_units):
            T.grad_sqnorm(u.g_shard, self.sqnorm, accumulate=i > 0)
        if dist.is_initialized():
            dist.all_reduce(self.sqnorm, op=dist.ReduceOp.SUM, group=self.group)
        for u in self.all_units:
            T.adamw_step(u.master, u.m, u.v, u.g_shard, u.w_shard, self.sqnorm, self.max_grad_norm, 1.0 / self.world, self.lr,
                         self.betas[0], self.betas[1], self.eps, u.decay, self.step_count)
            u.g_acc32 = None                   # the fp32 sum of a finished accumulation window is not kept across steps
        T.bump_weight_epoch()

    def grad_norm(self) -> torch.Tensor:
        return self.sqnorm.sqrt() / self.world

    def train_step(self, vision_x, lang_x, attention_mask=None, labels=None) -> torch.Tensor:
        self.zero_grad()
        out = self.model(vision_x, lang_x, attention_mask=attention_mask, labels=labels)
        loss = out[0] if not hasattr(out, "loss") else out.loss
        self.backward(loss)
        self.optimizer_step()
        return loss.detach()

    # ---- inspection ----------------------------------------------------------------------------------------------------
    def full_weights(self) -> torch.Tensor:
        """All trainable bf16 weights as one f32 vector in parameter order (gathers every unit once; tests, checkpoints)."""
        out = []
        for u in self.all_units:
            u.materialize()
            out.extend(p.detach().float().reshape(-1).cpu() for p in u.params)
            u.release()
        return torch.cat(out)

    def resident_bytes(self) -> int:
        """Bytes of weights, gradients and optimizer state this rank holds between steps."""
        return sum(u.n_shard * (2 + 2 + 12) for u in self.all_units)
