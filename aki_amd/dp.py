"""Data-parallel support for the AKI path: one process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI
on MI355X; "gloo" in the CPU tests).

Forward metric: samples are independent, so the batch is sharded over ranks and there is NO data-path collective
(`shard_batch`).  Training step (BASELINE configs[2]; reference: DDP at train/train.py:311-312, backward at
train/train_utils.py:252): the one real exchange is the gradient all-reduce, done here by `GradAllReducer`:
  * gradients are packed into a few LARGE flat buckets (default 512 MiB - xGMI is point-to-point, 7 links x ~153 GB/s
    per GPU, so ring collectives are per-link bound and want few, large messages rather than DDP's 25 MiB),
  * a bucket's all-reduce is launched asynchronously the moment its last gradient is produced by autograd
    (post-accumulate-grad hooks), overlapping the rest of the backward pass,
  * `finish()` waits, averages and leaves the result in every `p.grad`.
"""
from __future__ import annotations

from typing import Iterable, List, Optional

import torch
import torch.distributed as dist


def shard_batch(t: torch.Tensor, rank: Optional[int] = None, world: Optional[int] = None) -> torch.Tensor:
    """Contiguous batch shard of this rank (dim 0); the remainder goes to the first ranks."""
    rank = dist.get_rank() if rank is None else rank
    world = dist.get_world_size() if world is None else world
    n = t.shape[0]
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return t[lo: lo + base + (1 if rank < rem else 0)]


class _Bucket:
    def __init__(self, params: List[torch.nn.Parameter]):
        self.params = params
        self.numel = sum(p.numel() for p in params)
        self.pending = len(params)
        self.flat: Optional[torch.Tensor] = None
        self.work = None


class GradAllReducer:
    """Bucketed, backward-overlapped gradient averaging for a replica of the model.

        reducer = GradAllReducer(model.parameters())
        loss.backward()          # buckets are all-reduced as they fill
        reducer.finish()         # p.grad now holds the average over ranks
    """

    def __init__(self, params: Iterable[torch.nn.Parameter], bucket_bytes: int = 512 << 20, group=None,
                 reduce_dtype: Optional[torch.dtype] = None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.reduce_dtype = reduce_dtype
        ps = [p for p in params if p.requires_grad]
        ps.reverse()  # autograd produces gradients roughly in reverse registration order
        self.buckets: List[_Bucket] = []
        cur, cur_bytes, key = [], 0, None
        for p in ps:
            k = (p.dtype, p.device)
            nbytes = p.numel() * p.element_size()
            if cur and (k != key or cur_bytes + nbytes > bucket_bytes):
                self.buckets.append(_Bucket(cur))
                cur, cur_bytes = [], 0
            key = k
            cur.append(p)
            cur_bytes += nbytes
        if cur:
            self.buckets.append(_Bucket(cur))
        self._owner = {}
        self._hooks = []
        for b in self.buckets:
            for p in b.params:
                self._owner[p] = b
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))

    def _on_grad(self, p: torch.nn.Parameter):
        b = self._owner[p]
        b.pending -= 1
        if b.pending == 0:
            self._launch(b)

    def _launch(self, b: _Bucket):
        if self.world == 1:
            return
        dt = self.reduce_dtype or b.params[0].dtype
        b.flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1).to(dt) for p in b.params])
        b.work = dist.all_reduce(b.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def finish(self):
        for b in self.buckets:
            if self.world > 1:
                if b.work is None:       # a parameter got no gradient this step: reduce what we have
                    self._launch(b)
                b.work.wait()
                b.flat.div_(self.world)
                off = 0
                for p in b.params:
                    n = p.numel()
                    g = b.flat[off: off + n].view_as(p).to(p.dtype)
                    if p.grad is None:
                        p.grad = g.clone()
                    else:
                        p.grad.copy_(g)
                    off += n
            b.pending = len(b.params)
            b.flat = None
            b.work = None

    def remove(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []


class FlatGradReducer:
    """Gradient exchange over a FLAT gradient buffer (aki_amd/trainer.py): buckets are contiguous slices of the
    buffer, so nothing is packed or unpacked - RCCL reads and writes the gradients where the wgrad GEMMs put them.

    A bucket is launched (async, on the process group's own stream) as soon as every parameter that lives in it has been
    delivered by the backward pass (`notify`), overlapping the remaining backward; `finish()` launches whatever is left
    and waits.  Two modes:
      shard=False  all-reduce (DDP, train/train.py:311-312): every rank ends with the SUM of the whole bucket
      shard=True   in-place reduce-scatter (the FSDP / ZeRO exchange, train/distributed.py:170-243): rank r ends with
                   the SUM of sub-slice r of every bucket (`owned(b)`); the trainer updates only that slice and
                   `all_gather_weights` redistributes the bf16 weights - optimizer state is 1/world per rank
    The 1/world average is folded into the optimizer kernel's gradient scale.  Bucket size: xGMI is point-to-point
    (7 links x ~153 GB/s per GPU) and ring collectives are per-link bound, so few large messages (default 512 MiB)
    rather than DDP's 25 MiB.  Bucket boundaries also fall on `breaks` (optimizer segments) and, when sharding, bucket
    lengths are multiples of 8*world elements (the flat buffer is padded accordingly by the trainer).
    """

    def __init__(self, flat_grad: torch.Tensor, spans, bucket_bytes: int = 512 << 20, group=None, shard: bool = False,
                 breaks=(), exchange_when_alone: bool = False, flat32: Optional[torch.Tensor] = None,
                 first_bucket_bytes: Optional[int] = None):
        """spans: list of (param, start, stop) element ranges inside flat_grad, in buffer order.
        exchange_when_alone: issue the collectives even in a world of one rank (they are identities there) - how the RCCL
        entry points and the compute-stream -> communicator-stream hand-off are exercised on a single GPU.
        first_bucket_bytes: size of the FIRST bucket of every segment (None = bucket_bytes).  The buffer is laid out in forward order, the
        backward pass delivers it back to front: the first bucket (vision tokenizer, embeddings, first decoder layers) is the last one to be
        launched and its exchange is what the step waits for after the backward pass - a small one bounds that exposed tail
        (64 MiB: about 1 ms on one xGMI link) while the others stay large."""
        self.flat, self.group, self.shard = flat_grad, group, shard
        # fp32 exchange (the reference's DDP path under amp_bf16 all-reduces fp32 .grad, train/train.py:311-312): a bucket's slice of
        # `flat` is widened into `flat32` (same length) when the bucket is launched and the collective runs on the fp32 slice; the sum
        # stays there for the optimizer.  copy_in = False: flat32 already holds the gradients (an fp32 accumulation window).
        self.flat32, self.copy_in = flat32, True
        self.enabled = True           # False during the non-final micro-batches of a gradient-accumulation window
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.active = self.world > 1 or (exchange_when_alone and dist.is_initialized())
        # gloo (the CPU tests' transport) has neither reduce_scatter_tensor nor all_gather_into_tensor: it gets the same
        # result from all-reduces.  Chosen by BACKEND, never by catching exceptions - a failing RCCL call must surface.
        self.no_scatter = dist.is_initialized() and str(dist.get_backend(group)).lower() == "gloo"
        self._delivered = set()       # ids of parameters already counted in this accumulation window
        per = max(1, bucket_bytes // flat_grad.element_size())
        per_first = per if first_bucket_bytes is None else max(1, min(per, first_bucket_bytes // flat_grad.element_size()))
        brk = sorted(set(int(b) for b in breaks) | {flat_grad.numel()})
        self.buckets = []             # [start, stop, n_params, pending, work, widened into flat32]
        self._owner = {}
        self.stamps = None            # tools/train_bench.py: {bucket index: [event at launch, event at completion]} on the compute stream
        cur_start, cur_n, first = None, 0, True
        for i, (p, lo, hi) in enumerate(spans):
            if cur_start is None:
                cur_start = lo
            self._owner[id(p)] = len(self.buckets)
            cur_n += 1
            nxt = spans[i + 1][1] if i + 1 < len(spans) else flat_grad.numel()   # bucket ends where the next param starts
            if nxt - cur_start >= (per_first if first else per) or nxt in brk:
                self.buckets.append([cur_start, nxt, cur_n, cur_n, None, False])
                cur_start, cur_n, first = None, 0, nxt in brk
        if cur_start is not None:
            self.buckets.append([cur_start, flat_grad.numel(), cur_n, cur_n, None, False])
        if shard:
            for b in self.buckets:
                if (b[1] - b[0]) % (8 * self.world):
                    raise ValueError("sharded exchange needs bucket lengths that are multiples of 8*world elements")

    def owned(self, b):
        """(start, stop) of this rank's sub-slice of bucket b (the whole bucket when not sharding)."""
        if not self.shard:
            return b[0], b[1]
        c = (b[1] - b[0]) // self.world
        return b[0] + self.rank * c, b[0] + (self.rank + 1) * c

    def notify(self, p) -> None:
        """Called once a parameter's gradient slice is final for this window.  Idempotent per parameter: a weight used twice
        in one backward (tied / shared modules) delivers twice, and counting it twice would launch the bucket while other
        slices are still being written."""
        if not self.enabled or id(p) in self._delivered:
            return
        self._delivered.add(id(p))
        b = self.buckets[self._owner[id(p)]]
        b[3] -= 1
        if b[3] == 0:
            self._launch(b)

    def _launch(self, b) -> None:
        if b[4] is not None:
            return
        if self.flat32 is not None and self.copy_in and b[4] is None and not b[5]:
            self.flat32[b[0]: b[1]].copy_(self.flat[b[0]: b[1]])        # stream-ordered ahead of the collective
            b[5] = True
        if not self.active:
            return
        if self.stamps is not None and self.flat.is_cuda:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()                                                  # compute stream: the bucket's last gradient has been written
            self.stamps[self.buckets.index(b)] = [ev, None]
        src = self.flat if self.flat32 is None else self.flat32
        buf = src[b[0]: b[1]]
        if not self.shard or self.no_scatter:
            b[4] = dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            return
        lo, hi = self.owned(b)
        b[4] = dist.reduce_scatter_tensor(src[lo:hi], buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def finish(self) -> None:
        for b in self.buckets:
            self._launch(b)
        for i, b in enumerate(self.buckets):
            if b[4] is not None:
                b[4].wait()
                if self.stamps is not None and i in self.stamps and self.flat.is_cuda:
                    ev = torch.cuda.Event(enable_timing=True)
                    ev.record()                                          # compute stream, behind the wait: this bucket's sum is in place
                    self.stamps[i][1] = ev
            b[3], b[4], b[5] = b[2], None, False
        self._delivered.clear()

    def all_gather_weights(self, flat_w: torch.Tensor) -> None:
        """After the sharded optimizer step: every rank's freshly written sub-slices -> the full flat weight buffer."""
        if not self.shard or not self.active:
            return
        works = []
        for b in self.buckets:
            lo, hi = self.owned(b)
            if not self.no_scatter:
                works.append(dist.all_gather_into_tensor(flat_w[b[0]: b[1]], flat_w[lo:hi], group=self.group, async_op=True))
            else:                                          # gloo: zero the foreign slices and sum
                seg = flat_w[b[0]: b[1]]
                keep = flat_w[lo:hi].clone()
                seg.zero_()
                flat_w[lo:hi].copy_(keep)
                works.append(dist.all_reduce(seg, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        for w in works:
            w.wait()
