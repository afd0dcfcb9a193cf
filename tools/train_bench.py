"""Pre-training step of BASELINE configs[2] on real hardware: AKI-4B, bf16 compute / fp32 master weights, 8 samples per
GPU (336 px image + 512-token prompt, L = 655), forward + backward + gradient all-reduce (RCCL) + clip 1.0 + AdamW.
    python tools/train_bench.py [--gpus 1] [--steps 5] [--warmup 2] [--batch 8] [--layers 32] [--bucket-mb 512] [--shard-optimizer]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P tools/train_bench.py --gpus N ...
Prints ONE JSON line (rank 0) with the fields of bench.py's contract (value = whole-job training tokens/s over the
max-over-ranks time of exactly K steps bracketed by barrier + synchronize) plus the split into forward / backward (+ overlapped
exchange) / optimizer, `exchange_ms` = all gradient buckets exchanged back to back with nothing else running, `exchange_exposed_ms`
= what of it the step actually waits for after the backward pass, `overlap_frac` = 1 - exposed / total, model FLOP utilisation and
peak HBM use.  The gradients are exchanged in bf16 (they are produced in bf16 into the flat buffer; there is no fp32 gradient copy
to reduce), the sum is exact in fp32 inside RCCL's reduction only per element pair - see DESIGN.md section 6.
--exchange-when-alone runs the collectives in a world of one (identities through RCCL): the single-GPU box then measures the
software path of the exchange; real scaling needs the 8-GPU node (none has been available to this build: no scaling curve exists)."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def esz_of(tr):
    import torch
    return 4 if getattr(tr, "reduce_dtype", None) == torch.float32 else 2


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--layers", type=int, default=32)
    ap.add_argument("--shard-optimizer", action="store_true", help="ZeRO/FSDP-style: reduce-scatter grads, 1/world optimizer state (the default with more than one rank)")
    ap.add_argument("--no-emit-transposes", action="store_true", help="A/B: one aki_transpose per weight after every optimizer step instead of W^T out of the AdamW pass")
    ap.add_argument("--no-shard-optimizer", action="store_true", help="plain DDP replica: all-reduce, every rank updates everything")
    ap.add_argument("--reduce-dtype", choices=["bf16", "fp32"], default="bf16", help="fp32 = the reference's DDP arithmetic under amp_bf16 (twice the bytes)")
    ap.add_argument("--shard-params", action="store_true", help="FSDP FULL_SHARD equivalent: weights, gradients and optimizer state sharded (AkiShardedTrainer)")
    ap.add_argument("--gpus", type=int, default=1, help="ranks of the job; without a launcher (WORLD_SIZE unset) --gpus N > 1 starts the N ranks itself")
    ap.add_argument("--bucket-mb", type=int, default=512, help="gradient bucket size; xGMI is point-to-point, few large messages")
    ap.add_argument("--exchange-when-alone", action="store_true", help="world of one: still issue the RCCL collectives")
    ap.add_argument("--first-bucket-mb", type=int, default=64, help="size of the first bucket of each segment (its gradients finish last: the exposed tail); 0 = like the others")
    ap.add_argument("--no-fused-swiglu", action="store_true", help="A/B: gate_up and the SwiGLU as two launches in the training forward (rounds 1-5)")
    ap.add_argument("--lr-warmup", type=int, default=0, help="linear learning-rate warm-up over this many optimizer steps (the reference trains with one: train/train.py:127 "
                    "--warmup_steps 5000, 361-366); 0 = full rate from the first step, which is what the committed loss traces ran")
    ap.add_argument("--head-chunk", type=int, default=0, help="rows per chunk of the fused lm_head + cross-entropy (0 = default 2688: two chunks at the benchmark batch)")
    a = ap.parse_args()
    if a.no_fused_swiglu:
        from aki_amd.phi3 import Phi3MLP
        Phi3MLP.fuse_train_swiglu = False
    import bench
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:      # start the ranks ourselves, as a child process, before any GPU call
        raise SystemExit(bench.spawn_ranks(a.gpus, sys.argv[1:], script=os.path.abspath(__file__)))
    import torch
    import torch.distributed as dist
    rank, local_rank, world = (int(os.environ.get(k, d)) for k, d in (("RANK", "0"), ("LOCAL_RANK", "0"), ("WORLD_SIZE", "1")))
    if world != a.gpus:
        raise SystemExit(f"train_bench.py: --gpus {a.gpus} but the launcher started WORLD_SIZE={world} ranks")
    gloo = os.environ.get("AKI_BENCH_BACKEND", "nccl") != "nccl"      # test hook: ranks may share one GPU
    dev_index = local_rank % max(torch.cuda.device_count(), 1) if gloo else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1 or a.exchange_when_alone:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if gloo:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    ranks_seen = 1
    if dist.is_initialized():
        one = torch.ones(1, device="cpu" if gloo else dev)
        dist.all_reduce(one)
        ranks_seen = int(one.item())
    from aki_amd.factory import build_aki
    from aki_amd.trainer import AkiShardedTrainer, AkiTrainer
    from aki_amd.phi3 import make_phi3_config
    model = build_aki(make_phi3_config(num_hidden_layers=a.layers), dtype=torch.bfloat16, device=dev, seed=0)   # same seed: replicas start identical
    model.train()
    if a.head_chunk:
        model.lang_model.head_chunk_rows = a.head_chunk
    model.set_trainable()
    if a.shard_params:
        tr = AkiShardedTrainer(model, lr=1e-4, betas=(0.9, 0.999), weight_decay=0.01, max_grad_norm=1.0)
        tr.shard = True
        tr.reducer = type("R", (), {"finish": staticmethod(lambda: None), "active": False, "buckets": tr.all_units})()
    else:
        tr = AkiTrainer(model, lr=1e-4, betas=(0.9, 0.999), weight_decay=0.01, max_grad_norm=1.0, shard_optimizer=(True if a.shard_optimizer else False if a.no_shard_optimizer else None), reduce_dtype=(torch.float32 if a.reduce_dtype == "fp32" else None), emit_transposes=not a.no_emit_transposes,
                        bucket_bytes=a.bucket_mb << 20, exchange_when_alone=a.exchange_when_alone,
                        first_bucket_bytes=(a.first_bucket_mb << 20) if a.first_bucket_mb > 0 else None)
    B, L = a.batch, bench.N_TXT - 1 + bench.NV
    vx, ids, am = bench.synth_batch(B, dev, torch.bfloat16, model.media_token_id, seed=1000 + rank)
    labels = ids.clone()
    labels[labels == model.media_token_id] = -100          # train/losses.py:88-116: labels = input ids, special tokens masked
    ev = lambda: torch.cuda.Event(enable_timing=True)
    parts = {"forward": 0.0, "backward": 0.0, "optimizer": 0.0, "exchange_exposed": 0.0}
    # exposed exchange: from the end of the backward COMPUTE (autograd returned, last kernel queued) to the end of reducer.finish()
    orig_finish = tr.reducer.finish
    marks = {}
    def timed_finish():
        marks["a"] = ev(); marks["a"].record()
        orig_finish()
        marks["b"] = ev(); marks["b"].record()
    tr.reducer.finish = timed_finish
    losses = []
    base_lr = tr.lr
    bucket_rows = None
    torch.cuda.reset_peak_memory_stats()
    for it in range(a.warmup + a.steps):
        if it == a.warmup:
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            t0 = time.perf_counter()
        e = [ev() for _ in range(4)]
        last_step = it == a.warmup + a.steps - 1
        if last_step and getattr(tr.reducer, "active", False) and hasattr(tr.reducer, "stamps"):
            tr.reducer.stamps = {}                     # the last timed step also stamps every bucket (two event records per bucket)
        if a.lr_warmup > 0:
            tr.lr = base_lr * min(1.0, (it + 1) / a.lr_warmup)
        e[0].record()
        tr.zero_grad()
        out = model(vx, ids, attention_mask=am, labels=labels)
        e[1].record()
        tr.backward(out.loss)
        if a.shard_params and "a" not in marks:
            marks["a"] = marks["b"] = e[1]
        e[2].record()
        tr.optimizer_step()
        e[3].record()
        if it >= a.warmup:
            torch.cuda.synchronize()
            for k, i in (("forward", 0), ("backward", 1), ("optimizer", 2)):
                parts[k] += e[i].elapsed_time(e[i + 1])
            parts["exchange_exposed"] += marks["a"].elapsed_time(marks["b"])
            losses.append(float(out.loss))
            if last_step and getattr(tr.reducer, "stamps", None):
                # per bucket, on this rank's compute-stream clock, relative to the start of the backward pass: when its last gradient was written
                # (= when its collective was handed to the communicator stream) and when its sum was in place; the backward pass itself ends at
                # `backward_compute_end_ms`.  Buckets are in buffer (= forward) order: the backward pass launches them last to first.
                bucket_rows = {"backward_compute_end_ms": round(e[1].elapsed_time(marks["a"]), 3), "buckets": []}
                for i, b in enumerate(tr.reducer.buckets):
                    st = tr.reducer.stamps.get(i)
                    if st and st[0] is not None and st[1] is not None:
                        bucket_rows["buckets"].append({"index": i, "MiB": round((b[1] - b[0]) * esz_of(tr) / 2**20, 1), "launched_ms": round(e[1].elapsed_time(st[0]), 3),
                                                       "complete_ms": round(e[1].elapsed_time(st[1]), 3)})
                tr.reducer.stamps = None
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    elapsed_min = elapsed
    if world > 1:
        elapsed, elapsed_min = bench.rank_spread(elapsed, "cpu" if gloo else dev)
    # the whole exchange on its own: every bucket back to back, nothing else on the GPU
    exch_ms = 0.0
    if tr.reducer.active:
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t1 = time.perf_counter()
        for _ in range(3):
            tr.reducer.finish = orig_finish
            tr.reducer.finish()
            torch.cuda.synchronize()
        exch_ms = (time.perf_counter() - t1) / 3 * 1e3
    # bytes one rank hands to the collectives per step (the all-gather of weights under optimizer / parameter sharding included)
    exchange_dtype = str(getattr(tr, "reduce_dtype", torch.bfloat16)).replace("torch.", "")
    esz = 4 if exchange_dtype == "float32" else 2
    exchange_bytes = tr.numel * esz + (tr.numel * 2 if getattr(tr, "shard", False) else 0) + (2 * tr.numel * 2 if a.shard_params else 0)
    if rank == 0:
        n_lm = sum(p.numel() for n_, p in model.named_parameters() if n_.startswith("lang_model.") and "embed_tokens" not in n_)
        # 6 FLOP per parameter per token (fwd 2 + bwd 4) for the decoder + head, + the frozen tower's forward and the connector
        flops = 6.0 * n_lm * B * L
        ms = elapsed / a.steps * 1e3
        print(json.dumps({
            "metric": "training tokens/s, AKI-4B pre-training step (fwd+bwd+all-reduce+clip+AdamW)", "value": round(B * world * L * a.steps / elapsed, 1),
            "unit": "tokens/s", "n_gpus": world, "rccl_ranks": ranks_seen, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms, 2),
            "ms_per_step_rank_min": round(elapsed_min / a.steps * 1e3, 2), "ms_per_step_rank_max": round(ms, 2), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "AKI-4B pre-training step, BASELINE configs[2]: 336px image + 512-token prompt per sample, batch 8 per GPU, "
                                   "bf16 compute / fp32 master weights; random-init weights", "global_batch": B * world, "seq_len": L,
                       "parallelism": f"dp{world}" + ("+sharded-params" if a.shard_params else "+sharded-optimizer" if getattr(tr, "shard", False) else ""), "bucket_mb": a.bucket_mb,
                       "gradient_exchange": ("per-unit all-gather of weights (x2) + reduce-scatter of gradients" if a.shard_params else ("reduce-scatter + all-gather" if tr.shard else "all-reduce") + f" of {exchange_dtype} gradients, in place in the flat buffer"),
                       "layers": a.layers},
            "exchange_ms": round(exch_ms, 3), "exchange_exposed_ms": round(parts["exchange_exposed"] / a.steps, 3),
            "overlap_frac": (round(1.0 - min(1.0, (parts["exchange_exposed"] / a.steps) / exch_ms), 3) if exch_ms > 0 else None),
            "buckets": len(tr.reducer.buckets), "bucket_timeline": bucket_rows, "exchange_bytes": int(exchange_bytes), "exchange_dtype": exchange_dtype,
            "parts_ms": {k: round(v / a.steps, 2) for k, v in parts.items()}, "losses": [round(x, 4) for x in losses], "lr_warmup_steps": a.lr_warmup,
            "trainable_params": tr.numel, "lm_mfu_vs_2500TF": round(flops / (ms * 1e-3) / 2.5e15, 4),
            "peak_hbm_GB": round(torch.cuda.max_memory_allocated() / 1e9, 1), "precision": "bf16 compute, fp32 master/moments, bf16 grads", "head_chunk_rows": getattr(model.lang_model, "head_chunk_rows", 2688)}))
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
