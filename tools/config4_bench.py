"""BASELINE configs[3]: AKI-4B forward at seq = 4096 with 4 interleaved 336x336 images (multi-image MMA mask), bf16, one MI355X.
Per batch size: ms per forward, tokens/s, a `roofline` object for the dominant kernel (gate_up + SwiGLU, bracketed live with HIP events on
every 4th launch in a separate pass - the timed pass runs without probes) as in bench.py, and an `mma_core` object for the attention core
alone on the batch's own mask table (us, TF/s, fraction of the dense bf16 MFMA peak; `traffic` and `mfma_busy_frac` from the committed
rocprofv3 --pmc passes of tools/profile_attn64.sh when they were taken on this tree).   python tools/config4_bench.py > profiles/r06_config4_bench.json"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aki_amd import ops
import bench_legs
from aki_amd.factory import build_aki
dev = torch.device("cuda", 0)
model = build_aki(dtype=torch.bfloat16, device=dev).eval()
model.allow_multi_image = True
NV, N_IMG, L = 144, 4, 4096
N_TXT = L - N_IMG * (NV - 1)                      # 3524 prompt tokens incl. 4 placeholders
res = []
for B in (1, 2, 4):
    g = torch.Generator().manual_seed(B)
    ids = torch.randint(3, 31999, (B, N_TXT), generator=g)
    ids[:, 0] = 1
    for k, s in enumerate((6, 900 - 143, 1800 - 286, 2700 - 429)):   # placeholders so that the images start at 6 / 900 / 1800 / 2700 of the stream
        ids[:, s] = model.media_token_id
    ids[:, N_TXT - 64] = 32001
    vx = (torch.rand(B, N_IMG, 1, 3, 336, 336, generator=g) * 2 - 1).to(dev).to(torch.bfloat16)
    ids, am = ids.to(dev), torch.ones(B, N_TXT, dtype=torch.long, device=dev)
    with torch.no_grad():
        for _ in range(2):
            out = model(vx, ids, attention_mask=am)
        assert out.logits.shape[1] == L and torch.isfinite(out.logits.float()).all()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 5
        for _ in range(n):
            model(vx, ids, attention_mask=am)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
        tap = ops.EventTap(tags={"linear", "mma_attn_core", "mma_attn"}, every=4,
                           select=lambda tag: tag[0].startswith("mma_attn") or (tag[0] == "linear" and tag[4] == ops.ACT_SWIGLU))
        ops.set_event_tap(tap)
        for _ in range(2):
            model(vx, ids, attention_mask=am)
        summ = tap.summary()
        ops.set_event_tap(None)
        # the attention core alone: same shapes, the batch's own mask table
        prep = model._prepare_inputs_for_forward(vision_tokens=model.vision_tokenizer(model._encode_vision_x(vx)), lang_x=ids, attention_mask=am, padding_side="right")
        table = prep["attention_mask"]
        gq = torch.Generator(device=dev).manual_seed(7)
        q, k, v = (torch.randn(B, 32, L, 96, device=dev, generator=gq).to(torch.bfloat16) for _ in range(3))
        core_ms, core_sus_ms = bench_legs.core_times(ops, q, k, v, table)
        del q, k, v, prep
    pairs = L * (L + 1) // 2 + sum(NV * max(0, (L - 64) - (s_ + NV)) for s_ in (6, 900, 1800, 2700))
    cfl = 4.0 * 96 * pairs * 32 * B
    core = {"kernel": f"mma_attn64_bf16_kernel (64 rows per wave, one wave per SIMD) B{B} H32 L{L}, 4 images", "bound": "mfma", "us": round(core_ms * 1e3, 1), "us_sustained_40_launches": round(core_sus_ms * 1e3, 1), "timing": "best of 5 groups of 10 launches (as profiles/r06_attn_l4096_ab.txt)",
            "achieved": round(cfl / core_ms / 1e9, 1), "peak": 2500.0, "unit": "TFLOP/s", "frac": round(cfl / core_ms / 1e9 / 2500.0, 4), "mfma_frac": round(cfl / core_ms / 1e9 / 2500.0, 4),
            "algorithmic_flops_per_launch": cfl, "algorithmic_bytes_per_launch": int(4 * B * L * 3072 * 2), "traffic": None}
    pf = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", f"r06_attn64_b{B}_pmc.json")
    if os.path.exists(pf):
        from aki_amd.build import csrc_hash
        ent = json.load(open(pf))
        hit = [e_ for e_ in ent if "mma_attn64" in e_.get("kernel", "")]
        meta = [e_ for e_ in ent if e_.get("kernel") == "__meta__"]
        if hit:
            core.update(traffic=int(hit[0]["hbm_read_bytes_corrected_x2"] + hit[0]["hbm_write_bytes"]), traffic_unit="bytes per launch (L2<->fabric: FETCH_SIZE x2 + WRITE_SIZE; includes Infinity-Cache hits)",
                        traffic_source=os.path.relpath(pf), traffic_stale=bool(not meta or meta[0].get("csrc_tree_hash") != csrc_hash()),
                        mfma_busy_frac_of_simd_cycles=hit[0].get("mfma_busy_frac_of_simd_cycles"))
    gu = [(tag, n_, ms_) for tag, (n_, ms_) in summ.items() if tag[0] == "linear"]
    at = [(tag, n_, ms_) for tag, (n_, ms_) in summ.items() if tag[0].startswith("mma_attn")]
    (tag, _, gms), M = gu[0], B * L
    fl = 2.0 * M * 16384 * 3072
    roof = {"kernel": "gemm_bf16_kernel<8,4,2,4,SWIGLU> (gate_up + SwiGLU)", "bound": "mfma", "achieved": round(fl / gms / 1e9, 1), "peak": 2500.0, "unit": "TFLOP/s",
            "frac": round(fl / gms / 1e9 / 2500.0, 4), "traffic": None, "avg_launch_ms": round(gms, 4), "flop_per_launch": fl}
    # the whole MMA op as the decoder launches it (QKV projection with the RoPE epilogue + the core), timed inside the forward on every 4th launch
    op_ms = at[0][2] if at else None
    op_fl = 2.0 * M * 3072 * 9216 + cfl
    mma_op = None if not op_ms else {"kernel": "aki_mma_attn_fwd (QKV GEMM + RoPE epilogue, then the 64-row core), inside the forward", "bound": "mfma", "ms": round(op_ms, 4),
                                     "achieved": round(op_fl / op_ms / 1e9, 1), "peak": 2500.0, "unit": "TFLOP/s", "frac": round(op_fl / op_ms / 1e9 / 2500.0, 4), "flop_per_launch": op_fl}
    res.append({"batch": B, "seq_len": L, "images_per_sample": N_IMG, "ms_per_forward": round(ms, 2), "tokens_per_s": round(B * L / ms * 1e3, 1), "roofline": roof, "mma_core": core, "mma_op": mma_op,
                "mma_attention_ms_per_launch": {str(t_[0]): round(t_[2], 4) for t_ in at}})
print(json.dumps(res))
