"""Decode-path timing on one MI355X: full-size Phi-3.5-mini stream (random weights), MMA prefill into the KV cache,
then N greedy decode steps - eager launches vs hipGraph replay.  Prints ms/token and the weight-streaming rate
(every decoder + head weight is read once per token: the HBM roofline of this regime).
    python tools/decode_bench.py [--batch 1] [--prompt 655] [--steps 64]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--prompt", type=int, default=655)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--layers", type=int, default=32)
    ap.add_argument("--fp8", action="store_true", help="fp8 configuration: e4m3 prefill GEMMs, weight-only e4m3 GEMVs for batch 1")
    ap.add_argument("--norm-launch", action="store_true", help="batches of 2..8: the RMSNorm as a launch of its own instead of the skinny GEMM's prologue (A/B)")
    ap.add_argument("--batched-chain", action="store_true", help="batches of 2..8: the batched chain (opt-in) instead of five launches per layer")
    ap.add_argument("--no-chain", action="store_true", help="batch 1: the five-launch-per-layer path instead of the one-launch chain")
    a = ap.parse_args()
    from aki_amd import ops
    if a.batched_chain:                    # the batched chain is compiled into the lab library only (round 6)
        from aki_amd import _lib
        _lib._lib = _lib.load_lab()
    from aki_amd.phi3 import Phi3ForCausalLM, make_phi3_config, DecodeGraph
    from aki_amd.helpers import DecoupledEmbedding, DecoupledLinear
    dev = "cuda"
    torch.manual_seed(0)
    cfg = make_phi3_config(num_hidden_layers=a.layers)
    lm = Phi3ForCausalLM(cfg)
    for p in lm.parameters():
        p.data.normal_(0, 0.02)
    lm = lm.to(dev).to(torch.bfloat16).eval()
    if a.fp8:
        lm.enable_fp8()
    lm.model.use_decode_chain = not a.no_chain
    lm.model.use_decode_chain_batched = bool(a.batched_chain)
    ops.SKINNY_NORM_FUSED = not a.norm_launch
    wbytes = sum(p.numel() * 2 for n, p in lm.named_parameters() if "embed_tokens" not in n)
    B, L = a.batch, a.prompt
    x = torch.randn(B, L, cfg.hidden_size, device=dev, dtype=torch.bfloat16) * 0.5
    am = torch.ones(B, L, dtype=torch.bool)
    table = ops.MaskTable.from_host([[(4, 148, 4, 148)]] * B, am.numpy(), [L] * B, dev)
    if a.fp8 and B <= 16:
        wbytes //= 2
    # algorithmic HBM bytes of one step: every decoder + head weight once (B <= 16 rows ride on one pass), the K/V rows of the
    # cache once per sequence (mid-run length), activations negligible.  Peak: 8 TB/s (MI355X_MICROARCH.md; ~6.3 TB/s achievable).
    kv_bytes = B * a.layers * 2 * cfg.num_attention_heads * 96 * 2 * (L + 4 + a.steps // 2)
    res = {"batch": B, "prompt": L, "steps": a.steps, "weight_bytes": wbytes, "kv_bytes": kv_bytes, "fp8": bool(a.fp8),
           "chain": bool(not a.no_chain and (B == 1 or (B <= 8 and not a.fp8 and a.batched_chain)))}
    with torch.no_grad():
        for mode in ("eager", "graph"):
            out = lm(inputs_embeds=x, attention_mask=table, use_cache=True, cache_capacity=L + 2 * a.steps + 8)
            cache = out.past_key_values
            nxt = out.logits[:, -1].float().argmax(-1)
            stepper = DecodeGraph(lm, cache) if mode == "graph" else None
            step = (lambda ids: stepper.step(ids)) if stepper else (lambda ids: lm.decode_step(input_ids=ids, past_key_values=cache))
            for _ in range(4):
                nxt = step(nxt).float().argmax(-1)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.steps):
                nxt = step(nxt).float().argmax(-1)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) * 1e3 / a.steps
            res[mode] = {"ms_per_token": round(ms, 4), "tokens_per_s": round(B * 1e3 / ms, 1), "weight_GBps": round(wbytes / ms / 1e6, 1)}
            chain = getattr(cache, "chain", None)
            if chain is not None:
                chain.check()
        # what `generate` runs for greedy decoding: the pick (argmax, append, eos check, cache_len advance) INSIDE the replayed step;
        # checked against the plain graph's tokens
        out = lm(inputs_embeds=x, attention_mask=table, use_cache=True, cache_capacity=L + 2 * a.steps + 8)
        cache = out.past_key_values
        ref_cache = lm(inputs_embeds=x, attention_mask=table, use_cache=True, cache_capacity=L + 2 * a.steps + 8).past_key_values
        n_tok = 4 + a.steps + 1
        tokens = torch.full((B, n_tok), -1, dtype=torch.long, device=dev)
        pick = dict(pad_token_id=0, eos_ids=None, done=None, tokens=tokens, start_len=cache.cache_len.clone(), done_at=None)
        st = DecodeGraph(lm, cache, greedy=pick)
        ops.greedy_pick(out.logits[:, -1].contiguous(), st.ids, cache_len=cache.cache_len, advance=False, **pick)
        for _ in range(4):
            st.step_greedy()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            st.step_greedy()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3 / a.steps
        # the same token without a graph: five eager launches per token when the chain is on (the host runs ahead, nothing syncs)
        out_e = lm(inputs_embeds=x, attention_mask=table, use_cache=True, cache_capacity=L + 2 * a.steps + 8)
        cache_e = out_e.past_key_values
        tokens_e = torch.full((B, n_tok), -1, dtype=torch.long, device=dev)
        pick_e = dict(pad_token_id=0, eos_ids=None, done=None, tokens=tokens_e, start_len=cache_e.cache_len.clone(), done_at=None)
        ids_e = torch.zeros(B, dtype=torch.long, device=dev)
        ops.greedy_pick(out_e.logits[:, -1].contiguous(), ids_e, cache_len=cache_e.cache_len, advance=False, **pick_e)

        def eager_token():
            lg = lm.decode_step(input_ids=ids_e, past_key_values=cache_e, advance=False)
            ops.greedy_pick(lg, ids_e, cache_len=cache_e.cache_len, advance=True, **pick_e)
        for _ in range(4):
            eager_token()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            eager_token()
        torch.cuda.synchronize()
        ms_e = (time.perf_counter() - t0) * 1e3 / a.steps
        res["eager_greedy_pick"] = {"ms_per_token": round(ms_e, 4), "tokens_differing_from_the_graph": int((tokens_e[:, :9] != tokens[:, :9]).sum())}
        ref_st, nxt, want = DecodeGraph(lm, ref_cache), out.logits[:, -1].float().argmax(-1), []
        for _ in range(9):
            want.append(nxt)
            nxt = ref_st.step(nxt).float().argmax(-1)
        bad = int((torch.stack(want, 1) != tokens[:, :9]).sum())
        res["graph_greedy_pick_inside"] = {"ms_per_token": round(ms, 4), "tokens_per_s": round(B * 1e3 / ms, 1), "weight_GBps": round(wbytes / ms / 1e6, 1),
                                           "tokens_differing_from_the_plain_graph": bad}
        if getattr(cache, "chain", None) is not None:
            cache.chain.check()
    gbps = (wbytes + kv_bytes) / res["graph_greedy_pick_inside"]["ms_per_token"] / 1e6
    res["roofline"] = {"bound": "hbm", "achieved": round(gbps, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(gbps / 8000.0, 4),
                       "bytes_per_token": wbytes + kv_bytes, "timed": "hipGraph replay of one greedy token (step + pick), host wall clock over the timed steps"}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
