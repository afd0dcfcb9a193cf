#!/usr/bin/env python3
"""`AKI.generate` end to end, as the reference's callers run it (local_demo.py:76-87, eval_cv_bench/eval.py:99-104: one sample,
`generate(max_new_tokens=256)`, greedy): full AKI-4B (random-init), one 336 px image + 512-token prompt, batch 1.  Reports the time to the
first token (vision tower + connector + splice + MMA prefill into the KV cache) and the time per generated token (one hipGraph replay
each, the greedy pick inside it), with bf16 and with e4m3 weights.  No EOS (random weights never emit one on cue): all 256 tokens.
    python tools/generate_bench.py [--new 256] [--fp8] [--txt 64]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--new", type=int, default=256)
    ap.add_argument("--fp8", action="store_true")
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--txt", type=int, default=512, help="prompt tokens: 512 = the headline prompt (L = 655), 64 = BASELINE configs[0] (L = 207)")
    a = ap.parse_args()
    import bench
    bench.N_TXT = a.txt
    from aki_amd.factory import build_aki
    dev = torch.device("cuda", 0)
    model = build_aki(dtype=torch.bfloat16, device=dev, seed=0).eval()
    if a.fp8:
        model.lang_model.enable_fp8()
    vx, ids, am = bench.synth_batch(1, dev, torch.bfloat16, model.media_token_id, seed=1000)

    host = [0.0]

    def run(n_new):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        toks = model.generate(vx, ids, attention_mask=am, max_new_tokens=n_new, do_sample=False, eos_token_id=[])
        host[0] = time.perf_counter() - t0          # the call has returned; with one new token and no EOS set nothing in it synchronises
        torch.cuda.synchronize()
        return time.perf_counter() - t0, toks

    run(16)                                   # warm-up: allocator, lazily-built folds, the graph capture path
    res = {"prompt_tokens_lm_stream": bench.N_TXT - 1 + bench.NV, "new_tokens": a.new, "fp8": bool(a.fp8), "rounds": []}
    ref = None
    for _ in range(a.rounds):
        t1, _ = run(1)                        # prefill + first token
        h1 = host[0]
        tn, toks = run(a.new)
        assert toks.shape == (1, a.new)
        if ref is None:
            ref = toks.clone()
        assert torch.equal(ref, toks), "generate is not reproducible from call to call"
        res["rounds"].append({"first_token_ms": round(t1 * 1e3, 2), "first_token_host_issue_ms": round(h1 * 1e3, 2), "total_ms": round(tn * 1e3, 2),
                              "ms_per_new_token_after_the_first": round((tn - t1) * 1e3 / (a.new - 1), 4),
                              "new_tokens_per_s_end_to_end": round(a.new / tn, 1)})
    print(json.dumps(res))


if __name__ == "__main__":
    main()
