#!/usr/bin/env python3
"""Soak of the one-launch decode chain: the same greedy continuation twice (thousands of tokens each, eager launches and the in-graph pick),
full-size Phi-3.5-mini stream with random weights; the token sequences must be equal and the chain's sticky error word 0.  A hand-off
race (a consumer reading a vector before its producer's bytes) would show up as a diverging continuation.
    python tools/decode_chain_soak.py [--tokens 4096] [--fp8]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tokens", type=int, default=4096)
    ap.add_argument("--prompt", type=int, default=655)
    ap.add_argument("--fp8", action="store_true")
    a = ap.parse_args()
    from aki_amd import ops
    from aki_amd.phi3 import Phi3ForCausalLM, make_phi3_config, DecodeGraph
    cfg = make_phi3_config()
    lm = Phi3ForCausalLM(cfg)
    for p in lm.parameters():
        p.data.normal_(0, 0.02)
    lm = lm.to("cuda").to(torch.bfloat16).eval()
    if a.fp8:
        lm.enable_fp8()
    L, N = a.prompt, a.tokens
    x = torch.randn(1, L, cfg.hidden_size, device="cuda", dtype=torch.bfloat16) * 0.5
    table = ops.MaskTable.from_host([[(4, 148, 4, 148)]], torch.ones(1, L, dtype=torch.bool).numpy(), [L], "cuda")
    runs = []
    with torch.no_grad():
        for mode in ("eager", "graph", "eager"):
            out = lm(inputs_embeds=x, attention_mask=table, use_cache=True, cache_capacity=L + N + 8)
            cache = out.past_key_values
            tokens = torch.full((1, N), -1, dtype=torch.long, device="cuda")
            pick = dict(pad_token_id=0, eos_ids=None, done=None, tokens=tokens, start_len=cache.cache_len.clone(), done_at=None)
            ids = torch.zeros(1, dtype=torch.long, device="cuda")
            ops.greedy_pick(out.logits[:, -1].contiguous(), ids, cache_len=cache.cache_len, advance=False, **pick)
            st = None
            if mode == "graph":
                st = DecodeGraph(lm, cache, greedy=pick)
                st.ids.copy_(ids)
            t0 = time.perf_counter()
            for _ in range(1, N):
                if st is not None:
                    st.step_greedy()
                else:
                    lg = lm.decode_step(input_ids=ids, past_key_values=cache, advance=False)
                    ops.greedy_pick(lg, ids, cache_len=cache.cache_len, advance=True, **pick)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) * 1e3 / (N - 1)
            assert cache.chain is not None, "the chain is not active"
            cache.chain.check()
            runs.append((mode, tokens.clone(), round(ms, 4)))
    same = [bool(torch.equal(runs[0][1], r[1])) for r in runs[1:]]
    first_diff = [int((runs[0][1] != r[1]).nonzero()[0, 1]) if not torch.equal(runs[0][1], r[1]) else -1 for r in runs[1:]]
    print(json.dumps({"tokens": N, "prompt": L, "fp8": bool(a.fp8), "ms_per_token": {f"{m}_{i}": ms for i, (m, _, ms) in enumerate(runs)},
                      "continuations_equal_to_the_first": same, "first_differing_token": first_diff, "distinct_token_ids": int(runs[0][1].unique().numel())}))
    assert all(same), "the greedy continuation is not reproducible"


if __name__ == "__main__":
    main()
