"""Pre-training step of BASELINE configs[2] on real hardware: AKI-4B, bf16 compute / fp32 master weights, 8 samples per
GPU (336 px image + 512-token prompt, L = 655), forward + backward + gradient all-reduce (RCCL) + clip 1.0 + AdamW.
    python tools/train_bench.py [--steps 5] [--warmup 2] [--batch 8] [--layers 32]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 tools/train_bench.py
Prints one JSON line (rank 0): training tokens/s, ms per step, split into forward / backward(+all-reduce) / optimizer,
model FLOP utilisation against the dense bf16 MFMA peak, and peak HBM use."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--layers", type=int, default=32)
    ap.add_argument("--shard-optimizer", action="store_true", help="ZeRO/FSDP-style: reduce-scatter grads, 1/world optimizer state")
    a = ap.parse_args()
    import torch
    import torch.distributed as dist
    import bench
    rank, local_rank, world = (int(os.environ.get(k, d)) for k, d in (("RANK", "0"), ("LOCAL_RANK", "0"), ("WORLD_SIZE", "1")))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)
    from aki_amd.factory import build_aki
    from aki_amd.trainer import AkiTrainer
    from aki_amd.phi3 import make_phi3_config
    model = build_aki(make_phi3_config(num_hidden_layers=a.layers), dtype=torch.bfloat16, device=dev, seed=0)   # same seed: replicas start identical
    model.train()
    model.set_trainable()
    tr = AkiTrainer(model, lr=1e-4, betas=(0.9, 0.999), weight_decay=0.01, max_grad_norm=1.0, shard_optimizer=a.shard_optimizer)
    B, L = a.batch, bench.N_TXT - 1 + bench.NV
    vx, ids, am = bench.synth_batch(B, dev, torch.bfloat16, model.media_token_id, seed=1000 + rank)
    labels = ids.clone()
    labels[labels == model.media_token_id] = -100          # train/losses.py:88-116: labels = input ids, special tokens masked
    ev = lambda: torch.cuda.Event(enable_timing=True)
    parts = {"forward": 0.0, "backward": 0.0, "optimizer": 0.0}
    losses = []
    torch.cuda.reset_peak_memory_stats()
    for it in range(a.warmup + a.steps):
        if it == a.warmup:
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            t0 = time.perf_counter()
        e = [ev() for _ in range(4)]
        e[0].record()
        tr.zero_grad()
        out = model(vx, ids, attention_mask=am, labels=labels)
        e[1].record()
        tr.backward(out.loss)
        e[2].record()
        tr.optimizer_step()
        e[3].record()
        if it >= a.warmup:
            torch.cuda.synchronize()
            for k, i in (("forward", 0), ("backward", 1), ("optimizer", 2)):
                parts[k] += e[i].elapsed_time(e[i + 1])
            losses.append(float(out.loss))
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    if rank == 0:
        n_lm = sum(p.numel() for n_, p in model.named_parameters() if n_.startswith("lang_model.") and "embed_tokens" not in n_)
        # 6 FLOP per parameter per token (fwd 2 + bwd 4) for the decoder + head, + the frozen tower's forward and the connector
        flops = 6.0 * n_lm * B * L
        ms = elapsed / a.steps * 1e3
        print(json.dumps({
            "metric": "training tokens/s, AKI-4B pre-training step (fwd+bwd+all-reduce+clip+AdamW)", "value": round(B * world * L * a.steps / elapsed, 1),
            "unit": "tokens/s", "n_gpus": world, "ms_per_step": round(ms, 2), "global_batch": B * world, "seq_len": L,
            "parts_ms": {k: round(v / a.steps, 2) for k, v in parts.items()}, "losses": [round(x, 4) for x in losses],
            "trainable_params": tr.numel, "lm_mfu_vs_2500TF": round(flops / (ms * 1e-3) / 2.5e15, 4),
            "peak_hbm_GB": round(torch.cuda.max_memory_allocated() / 1e9, 1), "dtype": "bf16 compute, fp32 master/moments, bf16 grads"}))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
