#!/usr/bin/env python3
"""bench.py - image+text tokens/s of the AKI-4B forward pass on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = one forward pass of the whole AKI-4B model (SigLIP tower -> Perceiver connector -> splice ->
32 Phi-3.5-mini decoder layers with the fused MMA attention -> lm_head) over one batch of synthetic inputs that are
already resident in HBM: BASELINE.json configs[1], batch 8 per GPU, one 336x336 image + a 512-token chat prompt
per sample (LM-stream length L = 512 - 1 + 144 = 655 tokens/sample).  Random-init weights of the true architecture,
bf16.  Data parallel = independent replicas, batch sharded over ranks, no data-path collective (weak scaling);
the only collectives are the timing barrier and the max-over-ranks of the elapsed time.

Prints ONE JSON line (rank 0) with the contract's fields plus
  "roofline"     for the dominant kernel (by time in the timed region), timed live with HIP events on the launch stream
  "mma_kernel"   the same object for the north-star's MMA op (QKV projection + RoPE + span-driven attention)
  "mma_core"     the attention core alone (the op's second launch), timed live on the benchmark's own mask table and shapes
                 right after the timed region: algorithmic bytes / flops per launch against the HBM and MFMA peaks
  "cpu_baseline" the oracle (numpy port of the reference's eager path) timed on this box's host cores (N=1 only)
  "secondary"    (N=1 only, after the headline) the other BASELINE configurations and the decode / first-token / 384-px legs, each with
                 its own ms, steps, config.workload and roofline object - bench_legs.py
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0   # dense MFMA peak, MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0
N_TXT, NV, IMG_PX, BATCH = 512, 144, 336, 8


def synth_batch(B, device, dtype, media_id, seed):
    """[BOS, 5 system tokens, <image>, text ~U[3,31999], <|end|>=32007, <|assistant|>=32001 at N_txt-17,
    16 answer ids, EOS], no padding; images uniform[0,1) normalised to [-1,1)."""
    import torch
    g = torch.Generator(device="cpu").manual_seed(seed)
    ids = torch.randint(3, 32000, (B, N_TXT), generator=g)
    ids[:, 0] = 1
    ids[:, 6] = media_id
    ids[:, N_TXT - 18] = 32007
    ids[:, N_TXT - 17] = 32001
    ids[:, N_TXT - 1] = 2
    img = (torch.rand((B, 1, 1, 3, IMG_PX, IMG_PX), generator=g) - 0.5) / 0.5
    return img.to(device=device, dtype=dtype), ids.to(device), torch.ones_like(ids).to(device)


def _cpu_threads():
    """Thread count for the CPU legs.  The affinity mask can exceed what the box really grants (cgroup quota, SMT), and
    oversubscribed OpenMP teams are several times slower: pick the team size with the best measured GEMM rate, report it."""
    import torch
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    avail = cores
    xa, wa = torch.randn(2048, 3072), torch.randn(8192, 3072)
    best = (0.0, cores)
    for nt in sorted({c for c in (4, 8, 16, 32, 48, 64, 96, 128, 192, 256, avail) if c <= avail}):
        torch.set_num_threads(nt)
        torch.mm(xa, wa.t())
        t0 = time.perf_counter()
        for _ in range(3):
            torch.mm(xa, wa.t())
        rate = 3 * 2.0 * 2048 * 3072 * 8192 / (time.perf_counter() - t0)
        if rate > best[0] * 1.05:
            best = (rate, nt)
    torch.set_num_threads(best[1])
    return best[1], avail, best[0]


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def cpu_baseline(state_dict, media_id, threads=None):
    """The contract's CPU baseline: BASELINE configs[0] - batch 1, one 336 px image + a 64-token
    chat prompt (LM stream L = 207), fp32 - through the WHOLE eager torch restatement of the reference forward
    (oracle/aki_torch.py::aki_forward, pinned to the reference's own outputs: SigLIP 27 layers + Perceiver + splice + dense MMA
    mask + 4.41.2 inversion + 32 decoder layers + lm_head) on this box's host cores; 1 warm-up + 3 timed forwards, median.
    The weights are the benchmark model's own (random-init AKI-4B), converted to fp32 on the host (~17 GB)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import torch
    import aki_torch as OT
    cores, avail, rate = threads if threads is not None else _cpu_threads()
    n_txt = 64
    g = torch.Generator().manual_seed(0)
    ids = torch.randint(3, 32000, (1, n_txt), generator=g)
    ids[0, 0], ids[0, 6], ids[0, n_txt - 18], ids[0, n_txt - 17], ids[0, n_txt - 1] = 1, media_id, 32007, 32001, 2
    am = torch.ones_like(ids)
    vx = (torch.rand((1, 1, 1, 3, IMG_PX, IMG_PX), generator=g) - 0.5) / 0.5
    cfg = dict(vis_layers=27, vis_heads=16, lm_layers=32, lm_heads=32, max_original_id=32010, media_token_id=media_id,
               pad_token_id=32000, num_vision_tokens=NV)
    p32 = {k: v.detach().to("cpu", torch.float32) for k, v in state_dict.items()}
    L = n_txt - 1 + NV
    times = []
    with torch.no_grad():
        for i in range(4):
            t0 = time.perf_counter()
            out = OT.aki_forward(p32, cfg, vx, ids, am)["logits"]
            dt = time.perf_counter() - t0
            if i > 0:
                times.append(dt)
    assert out.shape[:2] == (1, L) and bool(torch.isfinite(out).all())
    del p32
    times.sort()
    med = times[len(times) // 2]
    return {"value": round(L / med, 2), "unit": "tokens/s", "cores": int(cores), "cores_visible": int(avail), "cpu_model": _cpu_model(),
            "cpu_gemm_gflops": round(rate / 1e9, 1), "kind": "port", "estimated": False,
            "sample": "BASELINE configs[0]: batch 1 x (336px image + 64-token prompt), L=207, fp32, the WHOLE forward of the eager torch "
                      "restatement of the reference (27 SigLIP + 6 Perceiver + 32 decoder layers + lm_head), 1 warm-up + 3 timed, median",
            "seconds_per_forward": round(med, 3), "seconds_all": [round(x, 3) for x in times], "tokens_per_forward": L}


def cpu_baseline_c2_est(threads=None):
    """Secondary figure (an ESTIMATE, kept next to the contract's baseline): the same restatement on the benchmark's own C2 shape
    (one batch of B samples, L = 655); 2 of 32 decoder layers and 2 of 27 SigLIP layers are timed (after one warm-up pass) and
    scaled by layer count - the layers are identical - while patch embed, Perceiver connector, splice + dense MMA mask + 4.41.2
    inversion and the lm_head are timed in full."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
    import torch
    import aki_torch as OT
    cores, avail, rate = threads if threads is not None else _cpu_threads()
    best = (rate, cores)
    cores = best[1]
    torch.set_num_threads(cores)
    B = BATCH
    g = torch.Generator().manual_seed(0)
    f = lambda *s: torch.randn(*s, generator=g) * 0.02
    ones, zeros = (lambda n: torch.ones(n)), (lambda n: torch.zeros(n))
    d, H, F, V = 3072, 32, 8192, 32064
    L = N_TXT - 1 + NV
    t = {}
    with torch.no_grad():
        # --- SigLIP: patch embed + 2 layers -----------------------------------------------------------------
        E, P, G = 1152, 14, IMG_PX // 14
        px = torch.rand(B, 3, IMG_PX, IMG_PX, generator=g) * 2 - 1
        wpe, bpe, pos = f(E, 3, P, P), f(E), f(G * G, E)
        OT.siglip_patch_embed(px, wpe, bpe, pos)
        t0 = time.perf_counter()
        h = OT.siglip_patch_embed(px, wpe, bpe, pos)
        t["patch_embed"] = time.perf_counter() - t0
        lp = {"layer_norm1.weight": ones(E), "layer_norm1.bias": zeros(E), "layer_norm2.weight": ones(E), "layer_norm2.bias": zeros(E),
              "mlp.fc1.weight": f(4304, E), "mlp.fc1.bias": f(4304), "mlp.fc2.weight": f(E, 4304), "mlp.fc2.bias": f(E)}
        for n_ in ("q", "k", "v", "out"):
            lp[f"self_attn.{n_}_proj.weight"] = f(E, E)
            lp[f"self_attn.{n_}_proj.bias"] = f(E)
        OT.siglip_encoder_layer(h, lp, 16)
        t0 = time.perf_counter()
        for _ in range(2):
            h = OT.siglip_encoder_layer(h, lp, 16)
        t["siglip_layer"] = (time.perf_counter() - t0) / 2
        # --- Perceiver connector (full) -------------------------------------------------------------------------
        pp = {"latents": torch.randn(NV, E, generator=g), "norm.weight": ones(E), "norm.bias": zeros(E),
              "projection.weight": f(d, E), "projection.bias": f(d)}
        for l in range(6):
            for nm in ("norm_media", "norm_latents"):
                pp[f"layers.{l}.0.{nm}.weight"] = ones(E)
                pp[f"layers.{l}.0.{nm}.bias"] = zeros(E)
            pp[f"layers.{l}.0.to_q.weight"] = f(512, E)
            pp[f"layers.{l}.0.to_kv.weight"] = f(1024, E)
            pp[f"layers.{l}.0.to_out.weight"] = f(E, 512)
            pp[f"layers.{l}.1.0.weight"] = ones(E)
            pp[f"layers.{l}.1.0.bias"] = zeros(E)
            pp[f"layers.{l}.1.1.weight"] = f(4 * E, E)
            pp[f"layers.{l}.1.3.weight"] = f(E, 4 * E)
        t0 = time.perf_counter()
        vt = OT.perceiver_resampler(h[:, None, None], pp)
        t["perceiver"] = time.perf_counter() - t0
        # --- splice + dense MMA mask + inversion (the reference materialises both) ------------------------------
        ids = torch.randint(3, 32000, (B, N_TXT), generator=g)
        ids[:, 6] = 32011
        ids[:, N_TXT - 17] = 32001
        emb = f(B, N_TXT, d)
        t0 = time.perf_counter()
        prep = OT.prepare_inputs_for_forward(vt, ids, torch.ones_like(ids), None, emb, 32011, 32000, NV)
        add = OT.invert_mask_441(prep["attention_mask"])
        t["splice_mask"] = time.perf_counter() - t0
        # --- 2 decoder layers ---------------------------------------------------------------------------------------
        dp = {"input_layernorm.weight": ones(d), "post_attention_layernorm.weight": ones(d),
              "self_attn.qkv_proj.weight": f(3 * d, d), "self_attn.o_proj.weight": f(d, d),
              "mlp.gate_up_proj.weight": f(2 * F, d), "mlp.down_proj.weight": f(d, F)}
        cos, sin = OT.rope_cos_sin(np.arange(L)[None], 96)
        x = prep["inputs_embeds"]
        OT.phi3_decoder_layer(x, dp, cos, sin, add, H)
        t0 = time.perf_counter()
        for _ in range(2):
            x = OT.phi3_decoder_layer(x, dp, cos, sin, add, H)
        t["decoder_layer"] = (time.perf_counter() - t0) / 2
        # --- final norm + lm_head -----------------------------------------------------------------------------------
        W, Wa = f(V, d), f(2, d)
        t0 = time.perf_counter()
        OT.decoupled_linear(OT.rms_norm(x, ones(d)), W, None, Wa, None, 32010)
        t["lm_head"] = time.perf_counter() - t0
    total = t["patch_embed"] + 27 * t["siglip_layer"] + t["perceiver"] + t["splice_mask"] + 32 * t["decoder_layer"] + t["lm_head"]
    return {"value": round(B * L / total, 2), "unit": "tokens/s", "cores": int(cores), "cores_visible": int(avail),
            "cpu_gemm_gflops": round(best[0] / 1e9, 1), "kind": "port", "estimated": True,
            "sample": f"one batch of the benchmark workload ({B} x (336px image + 512-token prompt), L=655), torch fp32 eager "
                      "restatement of the reference forward on the host cores (thread count = best measured GEMM rate): 2/32 decoder layers and 2/27 SigLIP layers "
                      "timed and scaled by layer count, patch embed, connector, splice + dense mask + inversion and lm_head "
                      "timed in full",
            "seconds_per_forward_est": round(total, 3), "parts_s": {k: round(v, 4) for k, v in t.items()}}


def rank_spread(elapsed, device):
    """(max, min) over the ranks of this rank's elapsed seconds - two scalar all-reduces after the timed region."""
    import torch
    import torch.distributed as dist
    hi = torch.tensor([elapsed], device=device, dtype=torch.float64)
    lo = hi.clone()
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    return float(hi.item()), float(lo.item())


def spawn_ranks(n, argv, script=None):
    """`python bench.py --gpus N` without a launcher: run `python -m torch.distributed.run --nproc-per-node N bench.py ...` as a
    CHILD process (nothing in this process has touched the GPU) and return its exit code.  Refuses when fewer than N GPUs are
    visible - unless the gloo test hook lets ranks share one.  (torch.cuda.device_count() counts through amdsmi where the build
    has it and through hipGetDeviceCount otherwise; either way the ranks are fresh child processes, never an exec of this one.)
    The launcher picks its own free rendezvous port (`--standalone`): no bind-close-reuse window for a concurrent run to take."""
    import subprocess
    import torch
    ndev = torch.cuda.device_count()
    if os.environ.get("AKI_BENCH_BACKEND", "nccl") == "nccl" and ndev < n:
        print(f"bench.py: --gpus {n} but only {ndev} GPUs are visible; not falling back to fewer ranks", file=sys.stderr)
        return 2
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--standalone", "--local-addr", "127.0.0.1",
           script or os.path.abspath(__file__)] + list(argv)
    return subprocess.run(cmd, env=env).returncode


def main():
    global IMG_PX
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=BATCH, help="per-GPU batch (BASELINE configs[1]: 8)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dtype", choices=["bf16", "fp8"], default="bf16",
                    help="fp8 = BASELINE configs[4]: e4m3 weights/activations for the decoder projections (use with --batch 16)")
    ap.add_argument("--kernel-table", action="store_true", help="also print the per-kernel timing table to stderr")
    ap.add_argument("--px", type=int, default=IMG_PX, choices=[336, 384],
                    help="image side of the synthetic batch: 336 = BASELINE's metric resolution (default), 384 = the tower's native size, what the "
                         "reference's SFT collate and demo feed (train/sft_data_utils/loader_utils.py:8, local_demo.py:20): 729 patches, M = 5832 rows in the tower at batch 8")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary legs (the other BASELINE configs, decode, first token, 384 px) that run after the headline at N = 1")
    args = ap.parse_args()
    IMG_PX = args.px

    # --gpus N outside a launcher: start the N ranks ourselves (one process per GPU) BEFORE anything touches the GPU, as a child
    # process whose exit code is ours; rank 0's JSON line goes to the inherited stdout.  Never falls back to one rank.
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args.gpus, sys.argv[1:]))

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback on the product path)")
    # AKI_BENCH_BACKEND=gloo is a test hook: several ranks may then share one GPU (the data path has no collective, only the
    # barrier and the max-over-ranks of the elapsed time go through the process group)
    backend = os.environ.get("AKI_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    if backend == "nccl" and world > ndev:
        raise SystemExit(f"bench.py: {world} ranks but only {ndev} GPUs are visible")
    dev_index = local_rank if backend == "nccl" else local_rank % max(ndev, 1)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    ranks_seen = 1
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
        # how many ranks the process group (RCCL on the GPU) really joins: an all-reduce of ones
        one = torch.ones(1, device=dev if backend == "nccl" else "cpu", dtype=torch.float32)
        dist.all_reduce(one)
        ranks_seen = int(one.item())
        assert ranks_seen == dist.get_world_size() == world, (ranks_seen, dist.get_world_size(), world)

    from aki_amd import ops
    from aki_amd.factory import build_aki
    B = args.batch
    model = build_aki(dtype=torch.bfloat16, device=dev, seed=rank)
    model.eval()
    fp8 = args.dtype == "fp8"
    if fp8:
        model.lang_model.enable_fp8()
    vx, ids, am = synth_batch(B, dev, torch.bfloat16, model.media_token_id, seed=1000 + rank)
    L = N_TXT - 1 + NV

    def step():
        with torch.no_grad():
            return model(vx, ids, attention_mask=am)

    for _ in range(args.warmup):
        out = step()
    assert out.logits.shape[:2] == (B, L)
    torch.cuda.synchronize()
    # HIP events on the launch stream around the kernels that are reported: the dominant GEMM (gate_up + SwiGLU, known
    # from profiles/) and the MMA op.  --kernel-table brackets every GEMM instead (costs ~4 % of the step).
    # An event record is not free (a barrier packet: ~6 us of dispatch gap each, tools/trace_gaps.py): the reported kernels
    # are bracketed on every 8th launch (4 of the 32 layers, every step: 40 samples per kernel in a default run), which keeps the probe
    # under 0.3 % of the step.
    tap = ops.EventTap(tags={"linear", "mma_attn", "linear_fp8", "mma_attn_fp8"}, every=1 if args.kernel_table else 8,
                       select=None if args.kernel_table else (lambda tag: tag[0].startswith("mma_attn") or
                                                              (tag[4] == ops.ACT_SWIGLU and tag[0] == ("linear_fp8" if fp8 else "linear"))))
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    ops.set_event_tap(tap)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    elapsed_min = elapsed
    ops.set_event_tap(None)
    if world > 1:
        dist.barrier()
        elapsed, elapsed_min = rank_spread(elapsed, dev if backend == "nccl" else "cpu")
    assert torch.isfinite(out.logits.float()).all()
    # the same loop once more WITHOUT the event probes (reported next to ms_per_step: what the probes cost)
    n_un = min(args.steps, 10)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n_un):
        out = step()
    torch.cuda.synchronize()
    ms_untapped = (time.perf_counter() - t0) / n_un * 1e3

    # the attention core on its own (second launch of the fused MMA op): same shapes, the batch's own mask table
    core = None
    if rank == 0:
        with torch.no_grad():
            prep = model._prepare_inputs_for_forward(vision_tokens=model.vision_tokenizer(model._encode_vision_x(vx)), lang_x=ids,
                                                     attention_mask=am, padding_side="right")
            table = prep["attention_mask"]
            gq = torch.Generator(device=dev).manual_seed(7)
            q, k, v = (torch.randn(B, 32, L, 96, device=dev, generator=gq).to(torch.bfloat16) for _ in range(3))
            for _ in range(3):
                ops.mma_attn_core(q, k, v, table, 96 ** -0.5)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                ops.mma_attn_core(q, k, v, table, 96 ** -0.5)
            e1.record()
            torch.cuda.synchronize()
            core = e0.elapsed_time(e1) / 20
            del q, k, v, prep

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        tokens = B * world * L
        value = tokens * args.steps / elapsed
        summ = tap.summary()
        rows = []
        for tag, (n_calls, avg_ms) in summ.items():
            if tag[0] in ("linear", "linear_fp8"):
                _, M, N, K, act = tag
                fl = 2.0 * M * N * K
                name = ("gemm_fp8" if tag[0] == "linear_fp8" else "gemm_bf16") + f" M{M} N{N} K{K}" + (" +swiglu" if act == 3 else "")
            else:
                _, b_, h_, l_, dh = tag
                pairs = l_ * (l_ + 1) // 2 + NV * max(0, (N_TXT - 17 + NV) - (6 + NV))
                fl = 2.0 * b_ * l_ * 3 * h_ * dh * (h_ * dh) + 4.0 * h_ * dh * pairs * b_
                name = ("mma_attn_fp8" if tag[0] == "mma_attn_fp8" else "mma_attn") + f" (qkv+rope+attention) B{b_} H{h_} L{l_}"
            rows.append(dict(kernel=name, tag=tag[0], calls_per_step=n_calls / args.steps, avg_ms=avg_ms,
                             total_ms_per_step=avg_ms * n_calls / args.steps, tflops=fl / avg_ms / 1e9, flops=fl))
        rows.sort(key=lambda r: -r["total_ms_per_step"])
        if args.kernel_table:
            for r in rows:
                print(f"  {r['kernel']:<48s} x{r['calls_per_step']:5.1f}/step  {r['avg_ms']:8.4f} ms  {r['tflops']:7.1f} TF/s  "
                      f"{r['total_ms_per_step']:8.3f} ms/step", file=sys.stderr)
        dom = rows[0]

        def pmc_traffic(*kernel_sigs):
            """HBM-side bytes per launch of one kernel (or the sum over the kernels of one op) from the committed rocprofv3
            --pmc passes of THIS command (profiles/*_pmc_summary.json: FETCH_SIZE with the gfx950 x2 correction + WRITE_SIZE;
            counters cannot be read from inside the process).  -> (bytes, source file, stale): stale = the summary was taken on
            another tree of aki_amd/csrc (or predates the hash stamp tools/summarize_prof.py writes).  (None, None, None) when no
            profile of the same configuration is committed."""
            if fp8 or world != 1 or B != BATCH:
                return None, None, None
            import glob
            from aki_amd.build import csrc_hash
            for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_summary.json")), reverse=True):
                try:
                    entries = json.load(open(f))
                    meta = [e for e in entries if e.get("kernel") == "__meta__"]
                    tot = 0
                    for sig in kernel_sigs:
                        hit = [e for e in entries if sig in e["kernel"] and "hbm_read_bytes_corrected_x2" in e and "hbm_write_bytes" in e]
                        if not hit:
                            raise KeyError(sig)
                        tot += int(hit[0]["hbm_read_bytes_corrected_x2"] + hit[0]["hbm_write_bytes"])
                    return tot, os.path.relpath(f, ROOT), (not meta) or meta[0].get("csrc_tree_hash") != csrc_hash()
                except Exception:
                    continue
            return None, None, None
        peak_of = lambda r: 5000.0 if r["tag"] == "linear_fp8" else PEAK_BF16_TFLOPS     # dense fp8 MFMA peak (guide): ~5 PF
        mk = lambda r: {"kernel": r["kernel"], "bound": "mfma", "achieved": round(r["tflops"], 1), "peak": peak_of(r),
                        "unit": "TFLOP/s", "frac": round(r["tflops"] / peak_of(r), 4), "traffic": None,
                        "avg_launch_ms": round(r["avg_ms"], 4), "algorithmic_flops_per_launch": r["flops"]}
        res = {
            "metric": f"image+text tokens/sec forward, AKI-4B, {IMG_PX}px img + 512 txt",
            "value": round(value, 1), "unit": "tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "ms_per_step_untapped": round(ms_untapped, 3), "rccl_ranks": ranks_seen,
            # every rank's own clock around the same K steps: `ms_per_step` is the slowest rank's (what `value` uses); a straggler shows here
            "ms_per_step_rank_min": round(elapsed_min / args.steps * 1e3, 3), "ms_per_step_rank_max": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "fp8-e4m3 projections (f32 accumulate), bf16 attention/residual" if fp8 else "bf16", "data": "synthetic",
            "config": {"workload": f"AKI-4B (Phi-3.5-mini + SigLIP-so400m/14 + Perceiver) forward, bf16, 1x{IMG_PX}px image + "
                                   f"512-token chat prompt per sample, batch {B} per GPU (BASELINE configs[{4 if fp8 else 1}]); random-init weights",
                       "global_batch": B * world, "seq_len": L, "tokens_per_sample": L, "patch_plus_text_tokens": (IMG_PX // 14) ** 2 + N_TXT,
                       "parallelism": f"dp{world}"},
            "roofline": mk(dom),
        }
        if dom["tag"] == "linear" and "+swiglu" in dom["kernel"]:
            tr, src, stale = pmc_traffic("gemm_bf16_kernel<8, 4, 2, 4, 1, 0")
            res["roofline"]["traffic"] = tr
            if src:
                res["roofline"]["traffic_stale"] = bool(stale)
                res["roofline"]["traffic_unit"] = "bytes per launch (L2<->fabric: FETCH_SIZE x2 + WRITE_SIZE; includes Infinity-Cache hits)"
                res["roofline"]["traffic_source"] = src
                res["roofline"]["algorithmic_bytes_per_launch"] = int(2 * (dom["flops"] / 2 / 16384 / 3072 * 3072 + 16384 * 3072 + dom["flops"] / 2 / 16384 / 3072 * 8192))
        mma = [r for r in rows if r["tag"].startswith("mma_attn")]
        TUNIT = "bytes per launch (L2<->fabric: FETCH_SIZE x2 + WRITE_SIZE; includes Infinity-Cache hits)"
        if mma:
            res["mma_kernel"] = mk(mma[0])
            tr, src, stale = pmc_traffic("gemm_bf16_kernel<8, 4, 2, 4, 3, 0", "mma_attn_bf16_kernel")
            if tr is not None:     # QKV+RoPE GEMM main launch + attention core (the 120-row tail launch of the GEMM is not in the sum)
                res["mma_kernel"].update(traffic=tr, traffic_unit=TUNIT, traffic_source=src, traffic_stale=bool(stale),
                                         algorithmic_bytes_per_launch=int(2 * (B * L * 3072 + 9216 * 3072 + 3 * B * L * 3072) + 4 * B * L * 3072 * 2))
        if core is not None:
            pairs = L * (L + 1) // 2 + NV * max(0, (N_TXT - 17 + NV) - (6 + NV))
            cfl, cby = 4.0 * 96 * pairs * B * 32, 4.0 * B * L * 3072 * 2
            tr, src, stale = pmc_traffic("mma_attn_bf16_kernel")
            res["mma_core"] = {"kernel": f"mma_attn_core (span-driven softmax(QK^T)V) B{B} H32 L{L}", "bound": "hbm",
                               "achieved": round(cby / core / 1e6, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                               "frac": round(cby / core / 1e6 / PEAK_HBM_GBS, 4), "traffic": tr, "avg_launch_ms": round(core, 4),
                               "algorithmic_bytes_per_launch": int(cby), "algorithmic_flops_per_launch": cfl,
                               "mfma_tflops": round(cfl / core / 1e9, 1), "mfma_frac": round(cfl / core / 1e9 / PEAK_BF16_TFLOPS, 4),
                               "note": "219 FLOP/B: just on the HBM side of the ridge (312 FLOP/B); both fractions are given"}
            if src:
                res["mma_core"].update(traffic_unit=TUNIT, traffic_source=src, traffic_stale=bool(stale))
        if world == 1 and not args.no_cpu_baseline and IMG_PX == 336:
            thr = _cpu_threads()
            res["cpu_baseline"] = cpu_baseline(model.state_dict(), model.media_token_id, thr)
            res["cpu_baseline"]["gpu_over_cpu"] = round(value / res["cpu_baseline"]["value"], 1)
            res["cpu_baseline_c2_est"] = cpu_baseline_c2_est(thr)
        # The other BASELINE configurations on the same clock, AFTER the headline's numbers exist: each leg is fenced (a failing leg
        # reports {"error": ...}), the headline values above are final, and the line is printed once.
        if world == 1 and not args.no_secondary and not fp8 and B == BATCH and IMG_PX == 336:
            try:
                import bench_legs
                res["secondary"] = bench_legs.run_all(model, dev, sys.modules[__name__])
            except Exception as e:      # noqa: BLE001
                res["secondary"] = {"error": f"{type(e).__name__}: {e}"[:300]}
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
