"""The one-launch decode step (decode_chain.hip) under the lab library's knobs, alternating settings in ONE process on one box:
poll period, copies of the hand-off vectors, READY flags per phase, and `nowait` (no dependency waits: wrong results, the time
of the bare weight stream in this workgroup structure).  hipGraph replay, host wall clock per token.
    python tools/decode_chain_ab.py [--prompt 655] [--steps 48] [--fp8]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--prompt", type=int, default=655)
    ap.add_argument("--steps", type=int, default=48)
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--fp8", action="store_true")
    a = ap.parse_args()
    from aki_amd import ops, _lib
    from aki_amd.phi3 import Phi3ForCausalLM, make_phi3_config, DecodeGraph
    dev = "cuda"
    torch.manual_seed(0)
    cfg = make_phi3_config()
    lm = Phi3ForCausalLM(cfg)
    for p in lm.parameters():
        p.data.normal_(0, 0.02)
    lm = lm.to(dev).to(torch.bfloat16).eval()
    if a.fp8:
        lm.enable_fp8()
    L = a.prompt
    x = torch.randn(1, L, cfg.hidden_size, device=dev, dtype=torch.bfloat16) * 0.5
    table = ops.MaskTable.from_host([[(4, 148, 4, 148)]], torch.ones(1, L, dtype=torch.bool).numpy(), [L], dev)
    # (s_sleep(1) x N per poll, copies of the hand-off vectors, READY flags per phase, nowait) on the product's batches
    settings = [("product: sleep 8, 1 copy, 32 flags", (8, 1, 32, 0)), ("sleep 2", (2, 1, 32, 0)), ("sleep 16", (16, 1, 32, 0)), ("2 copies", (8, 2, 32, 0)),
                ("4 copies", (8, 4, 32, 0)), ("16 flags", (8, 1, 16, 0)), ("one READY flag for the whole qkv phase (attention waits for all 576 producers)", (8, 1, 32, 0, 2)),
                ("nowait (wrong results): the bare weight stream", (8, 1, 32, 1))]
    res = {name: [] for name, _ in settings}
    with _lib.use_lab(0) as lab, torch.no_grad():
        for _ in range(a.rounds):
            for name, knobs in settings:
                lab.aki_lab_set_chain(*knobs[:4])
                lab.aki_lab_set_chain_nb(0)
                lab.aki_lab_set_chain_touch(knobs[4] if len(knobs) > 4 else -1)
                out = lm(inputs_embeds=x, attention_mask=table, use_cache=True, cache_capacity=L + a.steps + 16)
                cache = out.past_key_values
                nxt = out.logits[:, -1].float().argmax(-1)
                g = DecodeGraph(lm, cache)
                for _ in range(4):
                    nxt = g.step(nxt).float().argmax(-1)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(a.steps):
                    nxt = g.step(nxt).float().argmax(-1)
                torch.cuda.synchronize()
                res[name].append(round((time.perf_counter() - t0) * 1e3 / a.steps, 4))
                if not knobs[3]:
                    cache.chain.check()
        lab.aki_lab_set_chain(8, 1, 32, 0)
        lab.aki_lab_set_chain_nb(0)
        lab.aki_lab_set_chain_touch(-1)
    for name, _ in settings:
        print(f"{name:45s} ms/token {res[name]}")
    print(json.dumps(res))


if __name__ == "__main__":
    main()
