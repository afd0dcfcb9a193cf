"""The trainer's data-parallel path on real kernels: two ranks (sharing the one GPU of the test box, gloo standing in for RCCL)
train on halves of a batch - gradient buckets are exchanged from inside the HIP backward - and must end with the weights of a
single process training on the whole batch; both exchange modes (all-reduce, and reduce-scatter + sharded AdamW + all-gather)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _setup():
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    for p in (here, os.path.join(os.path.dirname(here), "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from test_train_gpu import _tiny_train_setup
    _, _, m, _, (vx, lx, am, lab) = _tiny_train_setup()
    # symmetric halves (rows 2,3 := rows 0,1): every rank sees the same number of valid targets, so the mean of the per-rank
    # mean losses is the full-batch mean loss
    vx, lx, am, lab = (torch.cat([t_[:2], t_[:2]], 0) for t_ in (vx, lx, am, lab))
    return m, vx, lx, am, lab


def _worker(rank, world, port, shard, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from aki_amd.trainer import AkiTrainer
    m, vx, lx, am, lab = _setup()
    tr = AkiTrainer(m, lr=2e-3, betas=(0.9, 0.95), weight_decay=0.1, bucket_bytes=1 << 20, shard_optimizer=shard)
    assert len(tr.reducer.buckets) >= 3 and tr.shard == shard
    sl = slice(2 * rank, 2 * rank + 2)
    losses = [float(tr.train_step(vx[sl], lx[sl], attention_mask=am[sl], labels=lab[sl])) for _ in range(2)]
    ret[rank] = (_weights_in_param_order(tr), losses, float(tr.grad_norm()))
    dist.barrier()
    dist.destroy_process_group()


def _weights_in_param_order(tr):
    """The bf16 weights as one vector in the trainer's parameter order (the flat buffers of a sharded and an unsharded
    trainer differ by alignment padding, the parameters do not)."""
    return torch.cat([p.detach().float().reshape(-1).cpu() for p in tr.params])


def _rccl_alone_worker(rank, port, shard, ret):
    """One rank, backend "nccl" (= RCCL): the collectives are identities but they are the REAL entry points -
    reduce_scatter_tensor / all_gather_into_tensor / all_reduce on slices that HIP kernels wrote on the compute stream."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    from aki_amd.trainer import AkiTrainer
    m, vx, lx, am, lab = _setup()
    tr = AkiTrainer(m, lr=2e-3, betas=(0.9, 0.95), weight_decay=0.1, bucket_bytes=1 << 20, shard_optimizer=shard, exchange_when_alone=True)
    assert tr.reducer.active and not tr.reducer.no_scatter and tr.shard == shard and len(tr.reducer.buckets) >= 3
    launched = []
    orig = tr.reducer._launch
    tr.reducer._launch = lambda b: (launched.append(torch.cuda.current_stream().query()), orig(b))[1]
    losses = [float(tr.train_step(vx, lx, attention_mask=am, labels=lab)) for _ in range(2)]
    torch.cuda.synchronize()
    ret["rccl"] = (_weights_in_param_order(tr), losses, float(tr.grad_norm()), len(launched))
    dist.destroy_process_group()


def _sharded_params_worker(rank, world, port, backend, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group(backend, rank=rank, world_size=world)
    from aki_amd.trainer import AkiShardedTrainer
    m, vx, lx, am, lab = _setup()
    n_train = sum(p.numel() for p in m.parameters() if p.requires_grad)
    tr = AkiShardedTrainer(m, lr=2e-3, betas=(0.9, 0.95), weight_decay=0.1)
    assert len(tr.units) == len(m.lang_model.model.layers) + 1 and len(tr.roots) == 2
    assert all(not u.live() for u in tr.all_units), "nothing but the shards is resident between steps"
    assert tr.resident_bytes() <= (n_train * 16) // world + 16 * 8 * world * len(tr.all_units)
    per = 4 // world
    sl = slice(per * rank, per * rank + per)
    losses = [float(tr.train_step(vx[sl], lx[sl], attention_mask=am[sl], labels=lab[sl])) for _ in range(2)]
    assert all(not u.live() for u in tr.all_units)
    with torch.no_grad():                                            # inference on released storages gathers what it needs
        out = m(vx[sl], lx[sl], attention_mask=am[sl])
    assert bool(torch.isfinite(out.logits).all()) and all(not u.live() for u in tr.all_units)
    ret[rank] = (tr.full_weights(), losses, float(tr.grad_norm()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(400)
@pytest.mark.parametrize("backend,world", [("gloo", 2), ("nccl", 1)], ids=["gloo-2-ranks", "rccl-1-rank"])
def test_parameter_sharded_training_matches_single_process(backend, world):
    """FSDP FULL_SHARD equivalent (AkiShardedTrainer): weights, gradients and optimizer state sharded over the ranks, units
    gathered around their forward / backward, gradients reduce-scattered per unit - against single-process AkiTrainer on
    the whole batch.  Two ranks share the one GPU over gloo (the logic, incl. unit-wise all-gather / reduce-scatter emulation);
    one rank over backend "nccl" runs the REAL all_gather_into_tensor / reduce_scatter_tensor entry points of RCCL."""
    from aki_amd.trainer import AkiTrainer
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_sharded_params_worker, args=(world, _free_port(), backend, ret), nprocs=world, join=True)
    for r in range(1, world):
        assert torch.equal(ret[0][0], ret[r][0]), "ranks disagree about the gathered weights"
    w, losses, gnorm = ret[0]
    m, vx, lx, am, lab = _setup()
    tr = AkiTrainer(m, lr=2e-3, betas=(0.9, 0.95), weight_decay=0.1)
    ref_losses = [float(tr.train_step(vx, lx, attention_mask=am, labels=lab)) for _ in range(2)]
    # the sharded trainer orders parameters unit by unit: compare as {name: tensor}
    ref = {n: p.detach().float().cpu() for n, p in m.named_parameters() if p.requires_grad}
    from aki_amd.trainer import AkiShardedTrainer
    m2, *_ = _setup()
    order = AkiShardedTrainer(m2, lr=0.0).params                      # same construction order, names from the module tree
    names = {id(p): n for n, p in m2.named_parameters()}
    off = 0
    worst, mean, cnt = 0.0, 0.0, 0
    for p in order:
        n = p.numel()
        d = (w[off:off + n] - ref[names[id(p)]].reshape(-1)).abs()
        worst, mean, cnt = max(worst, float(d.max())), mean + float(d.sum()), cnt + n
        off += n
    assert off == w.numel()
    mean_loss = sum(ret[r][1][1] for r in range(world)) / world
    assert abs(mean_loss - ref_losses[1]) < 2e-2 * abs(ref_losses[1])
    assert abs(gnorm - float(tr.grad_norm())) < 0.05 * float(tr.grad_norm()) + 1e-3
    assert mean / cnt < 2e-4 and worst <= 2 * 2 ** -7, (mean / cnt, worst)


@pytest.mark.timeout(300)
@pytest.mark.parametrize("shard", [False, True], ids=["allreduce", "sharded"])
def test_rccl_entry_points_on_one_gpu(shard):
    """The exchange through backend "nccl" (RCCL) with a world of one: same weights, bit for bit, as a trainer that runs
    no collective at all.  This is the stream hand-off gloo cannot test: RCCL reads the gradient slices on its own stream
    right after the wgrad GEMMs wrote them on the compute stream."""
    from aki_amd.trainer import AkiTrainer
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_rccl_alone_worker, args=(_free_port(), shard, ret), nprocs=1, join=True)
    w, losses, gnorm, n_launch = ret["rccl"]
    m, vx, lx, am, lab = _setup()
    tr = AkiTrainer(m, lr=2e-3, betas=(0.9, 0.95), weight_decay=0.1, bucket_bytes=1 << 20)   # same buckets -> same summation order of the norm
    ref_losses = [float(tr.train_step(vx, lx, attention_mask=am, labels=lab)) for _ in range(2)]
    assert n_launch >= 2 * 3, "the buckets were not exchanged"
    assert losses == ref_losses and gnorm == float(tr.grad_norm())
    assert torch.equal(w, _weights_in_param_order(tr))


@pytest.mark.timeout(300)
@pytest.mark.parametrize("shard", [False, True], ids=["allreduce", "sharded"])
def test_two_rank_training_matches_single_process(shard):
    from aki_amd.trainer import AkiTrainer
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), shard, ret), nprocs=world, join=True)
    (w0, l0, g0), (w1, l1, g1) = ret[0], ret[1]
    assert torch.equal(w0, w1), "replicas diverged"
    m, vx, lx, am, lab = _setup()
    tr = AkiTrainer(m, lr=2e-3, betas=(0.9, 0.95), weight_decay=0.1)
    ref_losses = [float(tr.train_step(vx, lx, attention_mask=am, labels=lab)) for _ in range(2)]
    wref = _weights_in_param_order(tr)
    assert abs(sum(l0) / 2 + sum(l1) / 2 - sum(ref_losses)) < 2e-2 * abs(sum(ref_losses))
    assert abs(g0 - float(tr.grad_norm())) < 0.05 * float(tr.grad_norm()) + 1e-3
    diff = (w0 - wref).abs()      # both exchange modes, compared in parameter order
    # the compared weights are the bf16 images: one ulp at |w| in [1, 2) is 2^-7
    assert float(diff.mean()) < 2e-4 and float(diff.max()) <= 2 * 2 ** -7, (float(diff.mean()), float(diff.max()))


def test_bench_contract_two_ranks_on_one_gpu():
    """bench.py launched the way the driver launches it for N > 1 (torch.distributed.run, one rank per GPU): here two
    ranks share the one GPU and gloo stands in for RCCL (AKI_BENCH_BACKEND test hook - the data path has no collective,
    only the barrier and the max-over-ranks of the elapsed time go through the process group).  Rank 0 prints exactly one
    JSON line that carries the contract's fields, with the whole-job token count of both ranks."""
    import json
    import subprocess
    import sys
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, AKI_BENCH_BACKEND="gloo")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--batch", "2", "--no-cpu-baseline"], capture_output=True, text=True, env=env, timeout=900, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["warmup"] == 1 and out["scaling"] == "weak" and out["higher_is_better"] is True
    assert out["unit"] == "tokens/s" and out["dtype"] == "bf16" and "workload" in out["config"] and "model" not in out["config"]
    # whole-job throughput: both ranks' tokens over the max-over-ranks time
    tokens = 2 * 2 * 655 * 2
    assert abs(out["value"] - tokens / (out["ms_per_step"] * 2 / 1e3)) / out["value"] < 1e-3
    assert out["roofline"]["bound"] in ("mfma", "hbm") and 0.0 < out["roofline"]["frac"] < 1.0
    # the day-one scaling table (tools/scale_report.py) from this line and a one-rank line of the same per-GPU batch
    r1 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--batch", "2", "--no-cpu-baseline",
                         "--no-secondary"], capture_output=True, text=True, timeout=900, cwd=root)
    assert r1.returncode == 0, r1.stderr[-2000:]
    one = [ln for ln in r1.stdout.splitlines() if ln.startswith("{")]
    assert len(one) == 1
    import tempfile
    with tempfile.NamedTemporaryFile("w", suffix=".jsonl", delete=False) as f:
        f.write(one[0] + "\n" + lines[0] + "\n")
    rep = subprocess.run([sys.executable, os.path.join(root, "tools", "scale_report.py"), f.name], capture_output=True, text=True, timeout=60)
    os.unlink(f.name)
    assert rep.returncode == 0, rep.stderr
    table = json.loads(rep.stdout)["forward"]
    assert [row["n_gpus"] for row in table] == [1, 2] and table[0]["x_scaling"] == 1.0
    assert 0.3 < table[1]["x_scaling"] < 2.2 and table[1]["rank_min_ms"] <= table[1]["rank_max_ms"]      # two ranks SHARE one GPU here: ~1x, not 2x


def test_bench_gpus_2_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it (the way the driver starts `--gpus 1`): bench.py must start the two
    ranks itself - as a child `torch.distributed.run` - and report n_gpus = 2 with both ranks in the process group; it must never
    run one rank and print a one-GPU line.  The gloo hook lets the two ranks share this box's single GPU."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["AKI_BENCH_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "2",
                        "--no-cpu-baseline"], capture_output=True, text=True, env=env, timeout=900, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["rccl_ranks"] == 2 and out["config"]["global_batch"] == 4 and out["config"]["parallelism"] == "dp2"
    # and with RCCL as the transport it refuses instead of falling back: this box has one GPU
    if torch.cuda.device_count() < 2:
        env.pop("AKI_BENCH_BACKEND")
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"],
                           capture_output=True, text=True, env=env, timeout=300, cwd=root)
        assert r.returncode != 0 and not [ln for ln in r.stdout.splitlines() if ln.startswith("{")], (r.returncode, r.stdout[-500:])


def _accum_worker(rank, world, port, shard, ret):
    """Each rank owns two samples and runs them as TWO micro-batches of one sample (gradient accumulation, fp32 sum), then one
    exchange + optimizer step."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from aki_amd.trainer import AkiTrainer
    m, vx, lx, am, lab = _setup()
    tr = AkiTrainer(m, lr=2e-3, betas=(0.9, 0.95), weight_decay=0.1, bucket_bytes=1 << 20, shard_optimizer=shard)
    launched = []
    orig = tr.reducer._launch
    tr.reducer._launch = lambda b: (launched.append(1), orig(b))[1]
    tr.zero_grad()
    for i in range(2):
        j = 2 * rank + i
        loss = m(vx[j:j + 1], lx[j:j + 1], attention_mask=am[j:j + 1], labels=lab[j:j + 1]).loss / 2
        n_before = len(launched)
        tr.backward(loss, last_microbatch=(i == 1))
        if i == 0:
            assert len(launched) == n_before, "a non-final micro-batch must not exchange anything"
    assert len(launched) >= 3
    tr.optimizer_step()
    ret[rank] = (_weights_in_param_order(tr), float(tr.grad_norm()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("shard", [False, True], ids=["allreduce", "sharded"])
def test_two_rank_gradient_accumulation_matches_single_process(shard):
    """Gradient accumulation under data parallelism: 2 ranks x 2 micro-batches of one sample (fp32 accumulation, the exchange only
    after the last micro-batch) against one process accumulating the same four samples - the loss weighting is the same on both
    sides (every micro-batch is one sample, loss / 2 per rank and 1 / world in the optimizer vs loss / 4), so the weights after the
    step agree to bf16 rounding of the exchanged sums."""
    from aki_amd.trainer import AkiTrainer
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_accum_worker, args=(world, _free_port(), shard, ret), nprocs=world, join=True)
    (w0, g0), (w1, g1) = ret[0], ret[1]
    assert torch.equal(w0, w1), "replicas diverged"
    m, vx, lx, am, lab = _setup()
    tr = AkiTrainer(m, lr=2e-3, betas=(0.9, 0.95), weight_decay=0.1)
    tr.zero_grad()
    for j in range(4):
        tr.backward(m(vx[j:j + 1], lx[j:j + 1], attention_mask=am[j:j + 1], labels=lab[j:j + 1]).loss / 4, last_microbatch=(j == 3))
    tr.optimizer_step()
    wref = _weights_in_param_order(tr)
    assert abs(g0 - float(tr.grad_norm())) < 0.03 * float(tr.grad_norm()) + 1e-3, (g0, float(tr.grad_norm()))
    diff = (w0 - wref).abs()
    assert float(diff.mean()) < 2e-4 and float(diff.max()) <= 2 * 2 ** -7, (float(diff.mean()), float(diff.max()))


def _fp32_exchange_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from aki_amd.trainer import AkiTrainer
    m, vx, lx, am, lab = _setup()
    default = AkiTrainer(m, lr=0.0)
    assert default.shard, "more than one rank: the optimizer state is sharded unless the caller asks for the plain replica"
    m, vx, lx, am, lab = _setup()
    tr = AkiTrainer(m, lr=2e-3, betas=(0.9, 0.95), weight_decay=0.1, bucket_bytes=1 << 20, shard_optimizer=False, reduce_dtype=torch.float32)
    assert tr.g32 is not None and not tr.shard and len(tr.reducer.buckets) >= 3
    sl = slice(0, 2) if rank == 0 else slice(1, 2)        # different samples per rank (_setup's halves are copies of each other)
    tr.zero_grad()
    out = m(vx[sl], lx[sl], attention_mask=am[sl], labels=lab[sl])
    tr.backward(out.loss)
    torch.cuda.synchronize()
    g_local, g_sum = tr.g16.float().cpu(), tr.g32.cpu()
    tr.optimizer_step()
    ret[rank] = (g_local, g_sum, _weights_in_param_order(tr), float(tr.grad_norm()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_fp32_gradient_exchange_two_ranks_is_the_exact_fp32_sum():
    """AkiTrainer(reduce_dtype=torch.float32): what the reference's DDP path exchanges under `--precision amp_bf16` (fp32 parameters,
    fp32 .grad, fp32 all-reduce: train/train.py:311-312, train/train_utils.py:56-65).  Two ranks (gloo, one GPU): after backward
    the fp32 buffer of both ranks holds exactly float(g_rank0) + float(g_rank1) of the bf16 gradients the HIP backward wrote, the
    bf16 buffers still hold the local gradients, and the optimizer (fp32-gradient AdamW) leaves both replicas with the same
    weights.  Also: with more than one rank the optimizer state is sharded by default."""
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_fp32_exchange_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    (g0, s0, w0, n0), (g1, s1, w1, n1) = ret[0], ret[1]
    assert torch.equal(s0, s1) and torch.equal(s0, g0 + g1)
    assert float((g0 - g1).abs().max()) > 0.0, "the ranks saw different samples"
    assert torch.equal(w0, w1) and n0 == n1
    assert abs(n0 - float((s0 / 2).double().norm())) < 1e-5 * max(1.0, n0)


def test_fp32_gradient_path_alone_equals_the_bf16_path():
    """One rank: the fp32 buffer is just the widened bf16 gradients, so sqnorm + AdamW from fp32 gradients reproduce the bf16-gradient
    step bit for bit - the two kernels differ in nothing but the load."""
    from aki_amd.trainer import AkiTrainer
    res = []
    for rd in (None, torch.float32):
        m, vx, lx, am, lab = _setup()
        tr = AkiTrainer(m, lr=2e-3, betas=(0.9, 0.95), weight_decay=0.1, bucket_bytes=1 << 20, reduce_dtype=rd)
        losses = [float(tr.train_step(vx, lx, attention_mask=am, labels=lab)) for _ in range(2)]
        res.append((_weights_in_param_order(tr), losses, float(tr.grad_norm())))
    assert res[0][1] == res[1][1] and res[0][2] == res[1][2]
    assert torch.equal(res[0][0], res[1][0])


def _default_sharded_fp32_worker(rank, world, port, shard, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from aki_amd.trainer import AkiTrainer
    m, vx, lx, am, lab = _setup()
    tr = AkiTrainer(m, lr=2e-3, betas=(0.9, 0.95), weight_decay=0.1, bucket_bytes=1 << 20, shard_optimizer=shard, reduce_dtype=torch.float32)
    assert tr.shard == shard and tr.g32 is not None
    sl = slice(0, 2) if rank == 0 else slice(1, 2)
    # ONE step: both layouts then start from identical weights and see identical fp32 gradient sums (from the second step on, a last-bit
    # difference in the bf16 weights feeds back through Adam's g / sqrt(v) on near-zero gradients and the runs drift apart legitimately)
    tr.zero_grad()
    tr.backward(m(vx[sl], lx[sl], attention_mask=am[sl], labels=lab[sl]).loss)
    tr.optimizer_step()
    full = tr.full_state_dict()
    own = tr.state_dict()
    # a consolidated state restores into a fresh trainer of the same layout and reproduces the weights
    m2, *_ = _setup()
    tr2 = AkiTrainer(m2, lr=2e-3, betas=(0.9, 0.95), weight_decay=0.1, bucket_bytes=1 << 20, shard_optimizer=shard, reduce_dtype=torch.float32)
    tr2.load_full_state_dict(full)
    # per parameter, in parameter order: the flat layouts of a sharded and a replicated trainer differ in their padding
    per_param = lambda flat: torch.cat([flat[tr.span_of[id(p)][0]: tr.span_of[id(p)][1]] for p in tr.params])
    ret[rank] = (_weights_in_param_order(tr), per_param(full["master"]), per_param(full["exp_avg"]), per_param(full["exp_avg_sq"]),
                 int(own["master"].numel()), _weights_in_param_order(tr2))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_default_multi_rank_combination_sharded_optimizer_with_fp32_exchange():
    """the default for world > 1 is a SHARDED optimizer, and with reduce_dtype=float32 that is a reduce-scatter on the fp32 buffer
    - a combination no test covered.  Two ranks, one step: the sharded run ends with the weights of the replicated (all-reduce) run (same
    fp32 sums, same AdamW arithmetic on every element, whoever owns it; the clip norm is summed in another order), its per-rank state_dict holds about half the state, and
    full_state_dict() - the consolidated state a checkpoint must hold - equals the replicated run's state and restores the same weights."""
    world = 2
    res = {}
    for shard in (True, False):
        mgr = mp.Manager()
        ret = mgr.dict()
        mp.spawn(_default_sharded_fp32_worker, args=(world, _free_port(), shard, ret), nprocs=world, join=True)
        res[shard] = (ret[0], ret[1])
    for shard in (True, False):
        a, b = res[shard]
        assert torch.equal(a[0], b[0]), "replicas diverged"
        for i in (1, 2, 3):
            assert torch.equal(a[i], b[i]), "full_state_dict differs between the ranks"
        assert torch.equal(a[5], a[0]), "load_full_state_dict did not restore the weights"
    sh, rep = res[True][0], res[False][0]
    # Same fp32 gradient sums and the same AdamW arithmetic per element; the one difference is the clip norm, which the sharded run adds up
    # per owner and all-reduces (another summation order: the clip factor differs in its last bits) - so: fp32 state equal to 1e-5, the bf16
    # weights equal except where that last bit crosses a rounding boundary (at most one bf16 step, on a small fraction of the elements).
    for i, name in ((1, "master"), (2, "exp_avg"), (3, "exp_avg_sq")):
        d = (sh[i] - rep[i]).abs()
        scale = rep[i].abs().mean()
        assert float(d.max()) <= 1e-4 * float(rep[i].abs().max()) and float(d.mean()) <= 2e-6 * float(scale), \
            f"consolidated {name} of the sharded run differs from the replicated run's: max {float(d.max()):.3g} mean {float(d.mean()):.3g} (mean |x| {float(scale):.3g})"
    dw = (sh[0].float() - rep[0].float()).abs()
    assert float((dw > 0).float().mean()) < 0.02 and bool((dw <= 2.0 ** -7 * rep[0].float().abs() + 1e-12).all()), \
        "sharded optimizer (reduce-scatter of fp32 sums) and replicated optimizer (all-reduce) disagree"
    assert sh[4] < 0.7 * rep[4] and res[True][0][4] + res[True][1][4] >= rep[4]
