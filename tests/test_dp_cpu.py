"""2-rank gloo tests of the data-parallel helpers (aki_amd/dp.py): batch sharding and the bucketed, backward-overlapped
gradient all-reduce against a single-process reference on the full batch."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _model():
    torch.manual_seed(0)
    return torch.nn.Sequential(torch.nn.Linear(16, 32), torch.nn.GELU(), torch.nn.Linear(32, 32, bias=False),
                               torch.nn.LayerNorm(32), torch.nn.Linear(32, 4))


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from aki_amd.dp import GradAllReducer, shard_batch
    m = _model()
    m[2].weight.requires_grad_(False)                      # a frozen parameter, like the vision tower
    red = GradAllReducer(m.parameters(), bucket_bytes=600)   # tiny buckets -> several async all-reduces
    assert len(red.buckets) >= 3
    g = torch.Generator().manual_seed(1)
    x = torch.randn(10, 16, generator=g)
    y = torch.randn(10, 4, generator=g)
    for step in range(2):                                  # two steps: bucket state must reset
        for p in m.parameters():
            p.grad = None
        xs, ys = shard_batch(x, rank, world), shard_batch(y, rank, world)
        loss = ((m(xs) - ys) ** 2).sum() / 10 * world      # so that the rank-average equals the full-batch gradient
        loss.backward()
        red.finish()
    ret[rank] = {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}
    dist.barrier()
    dist.destroy_process_group()


def test_shard_batch_covers_everything():
    from aki_amd.dp import shard_batch
    x = torch.arange(11)
    parts = [shard_batch(x, r, 4) for r in range(4)]
    assert torch.equal(torch.cat(parts), x) and [len(p) for p in parts] == [3, 3, 3, 2]


@pytest.mark.timeout(120)
def test_grad_all_reduce_world2_matches_full_batch():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    m = _model()
    m[2].weight.requires_grad_(False)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(10, 16, generator=g)
    y = torch.randn(10, 4, generator=g)
    (((m(x) - y) ** 2).sum() / 10).backward()
    want = {n: p.grad for n, p in m.named_parameters() if p.grad is not None}
    for r in range(world):
        assert set(ret[r].keys()) == set(want.keys())
        for n in want:
            torch.testing.assert_close(ret[r][n], want[n], atol=1e-6, rtol=1e-5)


def _flat_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from aki_amd.dp import FlatGradReducer
    # the trainer's layout: parameters are views of one flat buffer, 8-element aligned; gradients likewise
    shapes = [(16, 8), (5,), (32, 16), (3, 3), (64,)]
    spans, off = [], 0
    params = []
    for s in shapes:
        n = 1
        for d in s:
            n *= d
        p = torch.nn.Parameter(torch.zeros(s))
        spans.append((p, off, off + n))
        params.append(p)
        off = (off + n + 7) // 8 * 8
    flat = torch.zeros(off, dtype=torch.float32)
    red = FlatGradReducer(flat, spans, bucket_bytes=600 * 4 // 4, group=None)
    assert len(red.buckets) >= 2
    for step in range(2):                                   # twice: pending counters must re-arm
        flat.zero_()
        order = list(reversed(range(len(params))))          # backward order
        for i in order:
            p, lo, hi = spans[i]
            flat[lo:hi] = float(rank + 1) * (i + 1) + step
            red.notify(p)                                   # launches a bucket as soon as it is complete
        red.finish()
    ret[rank] = flat.clone()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_flat_grad_reducer_world2_sums_in_place():
    """FlatGradReducer (the trainer's gradient exchange): contiguous buckets of the flat buffer are all-reduced in place,
    launched from notify() in backward order; both ranks end with the SUM; alignment gaps stay zero."""
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_flat_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    a, b = ret[0], ret[1]
    assert torch.equal(a, b)
    shapes_n = [128, 5, 512, 9, 64]
    off = 0
    for i, n in enumerate(shapes_n):
        want = (1 + 2) * (i + 1) + 2 * 1                    # step 1 value: sum over ranks of (rank+1)*(i+1) + 1
        assert bool((a[off: off + n] == want).all()), (i, a[off: off + n][:4], want)
        nxt = (off + n + 7) // 8 * 8
        assert bool((a[off + n: nxt] == 0).all())
        off = nxt


def _shard_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from aki_amd.dp import FlatGradReducer
    align = 8 * world
    sizes = [130, 40, 500, 9, 64, 33]
    spans, off, params = [], 0, []
    per = 300                                               # elements per bucket (bucket_bytes = per * 4 for fp32)
    bstart = 0
    for n in sizes:
        p = torch.nn.Parameter(torch.zeros(n))
        spans.append((p, off, off + n))
        params.append(p)
        off = (off + n + 7) // 8 * 8
        if off - bstart >= per:
            off = (off + align - 1) // align * align
            bstart = off
    off = (off + align - 1) // align * align
    g = torch.zeros(off)
    w = torch.zeros(off)
    red = FlatGradReducer(g, spans, bucket_bytes=per * 4, group=None, shard=True, breaks=[off])
    assert len(red.buckets) >= 2 and all((b[1] - b[0]) % align == 0 for b in red.buckets)
    for step in range(2):
        g.zero_()
        for i in reversed(range(len(params))):
            p, lo, hi = spans[i]
            g[lo:hi] = float(rank + 1) * (i + 1) + step
            red.notify(p)
        red.finish()
        for b in red.buckets:                               # "optimizer": each rank updates only what it owns
            lo, hi = red.owned(b)
            w[lo:hi] = -g[lo:hi]
        red.all_gather_weights(w)
    ret[rank] = (w.clone(), [tuple(red.owned(b)) for b in red.buckets], [tuple(b[:2]) for b in red.buckets])
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_sharded_exchange_world2_reduce_scatter_then_all_gather():
    """shard=True (the FSDP / ZeRO exchange): every rank owns 1/world of each bucket, updates it, and the all-gather
    leaves identical full weights everywhere, equal to what the un-sharded all-reduce path would have produced."""
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_shard_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    (w0, own0, bk0), (w1, own1, bk1) = ret[0], ret[1]
    assert torch.equal(w0, w1) and bk0 == bk1
    for (lo0, hi0), (lo1, hi1), (s, e) in zip(own0, own1, bk0):
        assert (lo0, hi1) == (s, e) and hi0 == lo1            # the two ranks' slices tile each bucket
    sizes = [130, 40, 500, 9, 64, 33]
    # recompute the layout exactly as the worker did
    off, bstart, starts = 0, 0, []
    for n in sizes:
        starts.append(off)
        off = (off + n + 7) // 8 * 8
        if off - bstart >= 300:
            off = (off + 15) // 16 * 16
            bstart = off
    for i, (st, n) in enumerate(zip(starts, sizes)):
        want = -((1 + 2) * (i + 1) + 2 * 1)
        assert bool((w0[st: st + n] == want).all()), (i, w0[st: st + 4], want)


def _flat32_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from aki_amd.dp import FlatGradReducer
    n = 64
    p = torch.nn.Parameter(torch.zeros(n))
    q = torch.nn.Parameter(torch.zeros(n))
    spans = [(p, 0, n), (q, n, 2 * n)]
    flat = torch.zeros(2 * n, dtype=torch.bfloat16)
    flat32 = torch.full((2 * n,), 7.0, dtype=torch.float32)          # stale content must be overwritten, not added to
    red = FlatGradReducer(flat, spans, bucket_bytes=n * 2, group=None, flat32=flat32)
    assert len(red.buckets) == 2
    # rank 0 holds 1.0, rank 1 holds 1.5 * 2^-8: the exact sum 1.005859375 is not a bf16 number (a bf16 sum gives 1.0078125)
    flat.fill_(1.0 if rank == 0 else 1.5 * 2.0 ** -8)
    red.notify(q)
    red.notify(p)
    red.finish()
    first = flat32.clone()
    # an fp32 accumulation window: flat32 already holds the local sum, nothing is copied in
    flat32.fill_(float(rank + 1) + 2.0 ** -20)
    red.copy_in = False
    red.notify(q)
    red.notify(p)
    red.finish()
    ret[rank] = (first, flat32.clone(), flat.float().clone())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_flat_grad_reducer_fp32_exchange_world2():
    """AkiTrainer(reduce_dtype=torch.float32) - the reference's DDP arithmetic under amp_bf16 (fp32 gradients all-reduced in fp32,
    train/train.py:311-312): every bucket of the bf16 gradient buffer is widened into the fp32 buffer when it is launched, the
    collective sums the fp32 slices, the bf16 buffer is left as the backward wrote it.  The sum equals the exact fp32 sum of the
    ranks' bf16 gradients - a value a bf16 exchange cannot represent."""
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_flat32_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    for r in range(world):
        first, second, local = ret[r]
        assert bool((first == 1.0 + 1.5 * 2.0 ** -8).all())
        assert float(torch.tensor(1.0 + 1.5 * 2.0 ** -8).to(torch.bfloat16)) != 1.0 + 1.5 * 2.0 ** -8
        assert bool((second == 3.0 + 2.0 ** -19).all())
        assert bool((local == (1.0 if r == 0 else 1.5 * 2.0 ** -8)).all()), "the bf16 buffer must stay as the backward left it"


def test_first_bucket_of_each_segment_is_small():
    """The flat buffer is in forward order and the backward pass delivers it back to front, so the FIRST bucket of a segment is the last one
    launched - what the step waits for after the backward pass.  `first_bucket_bytes` keeps that one small; the others keep
    the large size, boundaries stay on parameter boundaries and on the segment breaks, and every parameter belongs to exactly one bucket."""
    from aki_amd.dp import FlatGradReducer
    sizes = [40, 40, 40, 400, 400, 400, 400, 24, 24, 24]                # two segments: seven weights, three norm gains
    params = [torch.nn.Parameter(torch.zeros(n)) for n in sizes]
    spans, off = [], 0
    for p in params[:7]:
        spans.append((p, off, off + p.numel()))
        off += p.numel()
    brk = off
    for p in params[7:]:
        spans.append((p, off, off + p.numel()))
        off += p.numel()
    flat = torch.zeros(off, dtype=torch.bfloat16)
    big = FlatGradReducer(flat, spans, bucket_bytes=800 * 2, breaks=[brk])
    small = FlatGradReducer(flat, spans, bucket_bytes=800 * 2, breaks=[brk], first_bucket_bytes=100 * 2)
    assert [tuple(b[:2]) for b in big.buckets] == [(0, 920), (920, 1720), (1720, 1792)]
    assert [tuple(b[:2]) for b in small.buckets] == [(0, 120), (120, 920), (920, 1720), (1720, 1792)]
    for red in (big, small):
        assert sum(b[2] for b in red.buckets) == len(params)
        assert all(red.buckets[red._owner[id(p)]][0] <= lo and hi <= red.buckets[red._owner[id(p)]][1] for p, lo, hi in spans)
