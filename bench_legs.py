"""Secondary legs of bench.py: the BASELINE.json configurations other than the headline, on the driver's clock.

bench.py (N = 1, rank 0) calls run_all() AFTER the headline region has been timed and its numbers computed; what comes back goes
under one "secondary" key of the same JSON line.  A leg that fails reports {"error": ...} and never takes the headline down.

  c3_l4096_forward    BASELINE configs[3]: seq 4096 with 4 interleaved 336-px images, batch 1, whole forward; plus `mma_core`, the
                      attention core alone on that mask (MFMA-bound at this length: fraction of the dense bf16 peak)
  px384_forward       the reference's native 384-px images (train/sft_data_utils/loader_utils.py:8, local_demo.py:20): 729 patches per
                      image through the tower, batch 8, L = 655
  decode_bf16         greedy decode, batch 1, one hipGraph replay per token (src/aki_generation.py:36-86): ms per token and the
                      weight-streaming rate against the HBM peak
  first_token         `generate(max_new_tokens=1)`, one sample (local_demo.py:75-87): L = 655 (headline prompt) and L = 207 (configs[0])
  c4_fp8_b16_forward  BASELINE configs[4]: e4m3 weights / activations in the decoder projections, batch 16, L = 655
  c2_train_step       BASELINE configs[2], the one-GPU leg: forward + backward + gradient exchange through RCCL (a world of one) +
                      clip + AdamW, batch 8

Every leg uses synthetic inputs already resident in HBM and random-init weights of the true architecture (bench.synth_batch)."""
import gc
import time

PEAK_BF16_TFLOPS = 2500.0
PEAK_FP8_TFLOPS = 5000.0
PEAK_HBM_GBS = 8000.0
NV = 144


def _timed(fn, warm, n):
    import torch
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, out


def _gate_up_roofline(ops, fn, M, fp8=False):
    """The dominant kernel of a forward (gate_up + SwiGLU), bracketed with HIP events on every 4th launch in a pass of its own."""
    tag0 = "linear_fp8" if fp8 else "linear"
    tap = ops.EventTap(tags={tag0}, every=4, select=lambda tag: tag[0] == tag0 and tag[4] == ops.ACT_SWIGLU)
    ops.set_event_tap(tap)
    try:
        fn()
        fn()
    finally:
        summ = tap.summary()
        ops.set_event_tap(None)
    rows = [(tag, n_, ms_) for tag, (n_, ms_) in summ.items()]
    if not rows:
        return None
    tag, _, ms = max(rows, key=lambda r: r[1] * r[2])
    fl = 2.0 * tag[1] * tag[2] * tag[3]
    peak = PEAK_FP8_TFLOPS if fp8 else PEAK_BF16_TFLOPS
    return {"kernel": ("gemm_fp8" if fp8 else "gemm_bf16") + f" M{tag[1]} N{tag[2]} K{tag[3]} +swiglu (gate_up)", "bound": "mfma", "achieved": round(fl / ms / 1e9, 1),
            "peak": peak, "unit": "TFLOP/s", "frac": round(fl / ms / 1e9 / peak, 4), "traffic": None, "avg_launch_ms": round(ms, 4)}


def core_times(ops, q, k, v, table):
    """The attention core alone: (best of 5 groups of 10 launches, mean of 40 launches in a row), ms.  The first is how
    profiles/r06_attn_l4096_ab.txt times its variants; the second is what a hot chip sustains (the clock gives way under this kernel:
    tools/attn64_stamps.py reads 1.8-2.0 GHz from inside it)."""
    import torch
    def run(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            ops.mma_attn_core(q, k, v, table, 96 ** -0.5)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n
    run(5)
    best = min(run(10) for _ in range(5))
    return best, run(40)


def leg_c3(model, dev, bench):
    import torch
    from aki_amd import ops
    N_IMG, L, B = 4, 4096, 1
    n_txt = L - N_IMG * (NV - 1)
    g = torch.Generator().manual_seed(3)
    ids = torch.randint(3, 31999, (B, n_txt), generator=g)
    ids[:, 0] = 1
    for s in (6, 900 - 143, 1800 - 286, 2700 - 429):      # placeholders: the images start at rows 6 / 900 / 1800 / 2700 of the LM stream
        ids[:, s] = model.media_token_id
    ids[:, n_txt - 64] = 32001
    vx = (torch.rand(B, N_IMG, 1, 3, 336, 336, generator=g) * 2 - 1).to(dev).to(torch.bfloat16)
    ids, am = ids.to(dev), torch.ones(B, n_txt, dtype=torch.long, device=dev)
    model.allow_multi_image = True
    try:
        with torch.no_grad():
            fn = lambda: model(vx, ids, attention_mask=am)
            ms, out = _timed(fn, 2, 5)
            assert out.logits.shape[1] == L and torch.isfinite(out.logits.float()).all()
            roof = _gate_up_roofline(ops, fn, B * L)
            # the attention core alone on this batch's own mask table
            prep = model._prepare_inputs_for_forward(vision_tokens=model.vision_tokenizer(model._encode_vision_x(vx)), lang_x=ids,
                                                     attention_mask=am, padding_side="right")
            table = prep["attention_mask"]
            gq = torch.Generator(device=dev).manual_seed(7)
            q, k, v = (torch.randn(B, 32, L, 96, device=dev, generator=gq).to(torch.bfloat16) for _ in range(3))
            core_ms, core_sus_ms = core_times(ops, q, k, v, table)
            del q, k, v, prep
    finally:
        model.allow_multi_image = False
    q_end = L - 64
    pairs = L * (L + 1) // 2 + sum(NV * max(0, q_end - (s + NV)) for s in (6, 900, 1800, 2700))
    cfl = 4.0 * 96 * pairs * 32 * B
    return {"ms": round(ms, 3), "steps": 5, "tokens_per_s": round(B * L / ms * 1e3, 1),
            "config": {"workload": "AKI-4B forward, bf16, seq 4096 with 4 interleaved 336-px images (BASELINE configs[3]), batch 1", "seq_len": L, "batch": B},
            "roofline": roof,
            "mma_core": {"kernel": f"mma_attn_core (64-row core) B{B} H32 L{L}, 4 images", "bound": "mfma", "achieved": round(cfl / core_ms / 1e9, 1), "peak": PEAK_BF16_TFLOPS,
                         "unit": "TFLOP/s", "frac": round(cfl / core_ms / 1e9 / PEAK_BF16_TFLOPS, 4), "mfma_frac": round(cfl / core_ms / 1e9 / PEAK_BF16_TFLOPS, 4),
                         "us": round(core_ms * 1e3, 1), "us_sustained_40_launches": round(core_sus_ms * 1e3, 1), "timing": "best of 5 groups of 10 launches (as profiles/r06_attn_l4096_ab.txt)", "traffic": None, "algorithmic_flops_per_launch": cfl, "visible_pairs_per_head": pairs}}


def leg_px384(model, dev, bench):
    import torch
    from aki_amd import ops
    B = 8
    g = torch.Generator(device="cpu").manual_seed(384)
    _, ids, am = bench.synth_batch(B, dev, torch.bfloat16, model.media_token_id, seed=1384)
    vx = ((torch.rand((B, 1, 1, 3, 384, 384), generator=g) - 0.5) / 0.5).to(device=dev, dtype=torch.bfloat16)
    L = bench.N_TXT - 1 + NV
    with torch.no_grad():
        fn = lambda: model(vx, ids, attention_mask=am)
        ms, out = _timed(fn, 2, 5)
        assert out.logits.shape[:2] == (B, L) and torch.isfinite(out.logits.float()).all()
        roof = _gate_up_roofline(ops, fn, B * L)
    return {"ms": round(ms, 3), "steps": 5, "tokens_per_s": round(B * L / ms * 1e3, 1), "patch_plus_text_tokens_per_s": round(B * (729 + bench.N_TXT) / ms * 1e3, 1),
            "config": {"workload": "AKI-4B forward, bf16, one 384-px image (the reference's native size: 729 patches, M = 5832 rows in the tower) + 512-token prompt, batch 8",
                       "seq_len": L, "batch": B}, "roofline": roof}


def leg_decode(model, dev, bench):
    import torch
    from aki_amd import ops
    from aki_amd.phi3 import DecodeGraph
    lm = model.lang_model
    cfg = lm.config
    B, L, steps = 1, 655, 64
    x = torch.randn(B, L, cfg.hidden_size, device=dev, dtype=torch.bfloat16) * 0.5
    table = ops.MaskTable.from_host([[(6, 150, 150, 638)]] * B, torch.ones(B, L, dtype=torch.bool).numpy(), [L] * B, dev)
    wbytes = sum(p.numel() * 2 for n, p in lm.named_parameters() if "embed_tokens" not in n)
    kv_bytes = B * cfg.num_hidden_layers * 2 * cfg.num_attention_heads * 96 * 2 * (L + 4 + steps // 2)
    with torch.no_grad():
        out = lm(inputs_embeds=x, attention_mask=table, use_cache=True, cache_capacity=L + 2 * steps + 8)
        cache = out.past_key_values
        n_tok = 4 + steps + 1
        tokens = torch.full((B, n_tok), -1, dtype=torch.long, device=dev)
        pick = dict(pad_token_id=0, eos_ids=None, done=None, tokens=tokens, start_len=cache.cache_len.clone(), done_at=None)
        st = DecodeGraph(lm, cache, greedy=pick)
        ops.greedy_pick(out.logits[:, -1].contiguous(), st.ids, cache_len=cache.cache_len, advance=False, **pick)
        for _ in range(4):
            st.step_greedy()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            st.step_greedy()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3 / steps
        chain = getattr(cache, "chain", None)
        if chain is not None:
            chain.check()
        del st, cache, out
    gbps = (wbytes + kv_bytes) / ms / 1e6
    return {"ms": round(ms, 4), "steps": steps, "tokens_per_s": round(B * 1e3 / ms, 1),
            "config": {"workload": "greedy decode, batch 1, prompt of 655 rows in the KV cache, one hipGraph replay per token (32-layer dataflow launch + head + pick)", "batch": B, "prompt": L},
            "roofline": {"kernel": "decode token (every decoder + head weight once, K/V rows once)", "bound": "hbm", "achieved": round(gbps, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                         "frac": round(gbps / PEAK_HBM_GBS, 4), "traffic": None, "algorithmic_bytes_per_launch": int(wbytes + kv_bytes)}}


def leg_first_token(model, dev, bench):
    import torch
    res = {}
    saved = bench.N_TXT
    try:
        for n_txt in (512, 64):
            bench.N_TXT = n_txt
            vx, ids, am = bench.synth_batch(1, dev, torch.bfloat16, model.media_token_id, seed=1000)

            def run():
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                model.generate(vx, ids, attention_mask=am, max_new_tokens=1, do_sample=False, eos_token_id=[])
                torch.cuda.synchronize()
                return (time.perf_counter() - t0) * 1e3
            run(); run()
            ts = [run() for _ in range(5)]
            res[f"L{n_txt - 1 + NV}"] = {"ms": round(min(ts), 3), "ms_median": round(sorted(ts)[2], 3), "steps": 5}
    finally:
        bench.N_TXT = saved
    res["config"] = {"workload": "generate(max_new_tokens=1), one sample: vision tower + connector + splice + MMA prefill into the KV cache + first greedy token; "
                                 "L655 = the headline prompt, L207 = BASELINE configs[0]'s (64-token prompt)"}
    res["ms"] = res["L655"]["ms"]
    res["steps"] = 5
    return res


def leg_fp8(model, dev, bench):
    import torch
    from aki_amd import ops
    B = 16
    model.lang_model.enable_fp8()
    vx, ids, am = bench.synth_batch(B, dev, torch.bfloat16, model.media_token_id, seed=1016)
    L = bench.N_TXT - 1 + NV
    with torch.no_grad():
        fn = lambda: model(vx, ids, attention_mask=am)
        ms, out = _timed(fn, 2, 5)
        assert out.logits.shape[:2] == (B, L) and torch.isfinite(out.logits.float()).all()
        roof = _gate_up_roofline(ops, fn, B * L, fp8=True)
    return {"ms": round(ms, 3), "steps": 5, "tokens_per_s": round(B * L / ms * 1e3, 1),
            "dtype": "fp8-e4m3 projections (f32 accumulate), bf16 attention/residual",
            "config": {"workload": "AKI-4B forward, e4m3 weights/activations in the decoder projections, 336-px image + 512-token prompt, batch 16 (BASELINE configs[4])",
                       "seq_len": L, "batch": B}, "roofline": roof}


def leg_train(dev, bench):
    import os
    import torch
    import torch.distributed as dist
    from aki_amd.factory import build_aki
    from aki_amd.trainer import AkiTrainer
    exchange = False
    try:                                   # a world of one through RCCL: the software path of the gradient exchange on the step
        if not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29531")
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        exchange = True
    except Exception:                      # noqa: BLE001
        exchange = False
    model = build_aki(dtype=torch.bfloat16, device=dev, seed=0)
    model.train()
    model.set_trainable()
    tr = AkiTrainer(model, lr=1e-4, betas=(0.9, 0.999), weight_decay=0.01, max_grad_norm=1.0, shard_optimizer=False, bucket_bytes=512 << 20,
                    exchange_when_alone=exchange, first_bucket_bytes=64 << 20)
    B, L = 8, bench.N_TXT - 1 + NV
    vx, ids, am = bench.synth_batch(B, dev, torch.bfloat16, model.media_token_id, seed=1000)
    labels = ids.clone()
    labels[labels == model.media_token_id] = -100
    losses = []
    parts = [0.0, 0.0, 0.0]
    steps, warm = 5, 2
    ev = lambda: torch.cuda.Event(enable_timing=True)
    for it in range(warm + steps):
        if it == warm:
            torch.cuda.synchronize()
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
        if it >= warm:
            torch.cuda.synchronize()
            for i in range(3):
                parts[i] += e[i].elapsed_time(e[i + 1])
            losses.append(float(out.loss.detach()))
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    n_lm = sum(p.numel() for n_, p in model.named_parameters() if n_.startswith("lang_model.") and "embed_tokens" not in n_)
    flops = 6.0 * n_lm * B * L
    res = {"ms": round(ms, 2), "steps": steps, "tokens_per_s": round(B * L / ms * 1e3, 1),
           "parts_ms": {"forward": round(parts[0] / steps, 2), "backward": round(parts[1] / steps, 2), "optimizer": round(parts[2] / steps, 2)},
           "losses": [round(x, 4) for x in losses], "exchange_when_alone": bool(exchange),
           "config": {"workload": "AKI-4B pre-training step (BASELINE configs[2], the one-GPU leg): forward + backward + gradient exchange (RCCL, world of one) + clip 1.0 + AdamW, "
                                  "bf16 compute / fp32 master weights, batch 8, L = 655", "batch": B, "seq_len": L},
           "roofline": {"kernel": "whole step, decoder + head FLOPs (6 per parameter per token)", "bound": "mfma", "achieved": round(flops / (ms * 1e-3) / 1e12, 1), "peak": PEAK_BF16_TFLOPS,
                        "unit": "TFLOP/s", "frac": round(flops / (ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4), "traffic": None}}
    del tr, model, out
    if exchange and dist.is_initialized():
        try:
            dist.destroy_process_group()
        except Exception:                  # noqa: BLE001
            pass
    return res


def run_all(model, dev, bench, budget_s=200.0):
    """-> dict of legs.  `model` is the headline's model (bf16, eval); the fp8 leg converts its decoder weights, so the caller must not
    use it for bf16 work afterwards.  The training leg builds its own model."""
    import torch
    t_start = time.perf_counter()
    out = {}

    def leg(name, fn, *a):
        if time.perf_counter() - t_start > budget_s:
            out[name] = {"error": f"skipped: the secondary legs' time budget ({budget_s:.0f} s) was spent"}
            return
        t0 = time.perf_counter()
        try:
            out[name] = fn(*a)
        except Exception as e:             # noqa: BLE001
            out[name] = {"error": f"{type(e).__name__}: {e}"[:300]}
        out[name]["leg_wall_s"] = round(time.perf_counter() - t0, 1)
        gc.collect()
        torch.cuda.empty_cache()

    leg("c3_l4096_forward", leg_c3, model, dev, bench)
    leg("px384_forward", leg_px384, model, dev, bench)
    leg("decode_bf16", leg_decode, model, dev, bench)
    leg("first_token", leg_first_token, model, dev, bench)
    leg("c4_fp8_b16_forward", leg_fp8, model, dev, bench)          # converts the model's decoder weights: after every bf16 leg
    leg("c2_train_step", leg_train_child)                          # a child process with its own (trainable) model; 288 GB of HBM hold both
    return out


def leg_train_child():
    """The training leg in a CHILD process: RCCL prints a version banner on stdout when its communicator comes up (bench.py's stdout is one
    JSON line), and a crash in there must not reach the headline.  The child is started fresh - nothing of this process is exec'ed over."""
    import json
    import os
    import subprocess
    import sys
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--train-leg"], capture_output=True, text=True, timeout=170, env=env)
    for line in reversed(r.stdout.strip().split("\n")):
        if line.startswith("{"):
            return json.loads(line)
    return {"error": f"training leg: rc {r.returncode}: {(r.stderr or r.stdout)[-300:]}"}


if __name__ == "__main__":
    import json
    import os
    import sys
    if "--train-leg" in sys.argv:
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        import torch
        import bench
        torch.cuda.set_device(0)
        print(json.dumps(leg_train(torch.device("cuda", 0), bench)), flush=True)
