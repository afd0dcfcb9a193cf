"""The one-launch decode step against the number of workgroups resident per CU (lab library: unused dynamic LDS caps it) and the
batches per workgroup - hypothesis: a hop of the dependency chain (poll, x load, arrival) queues behind the weight loads the SAME CU has
in flight (MI355X_MICROARCH.md, handoff-1to1: 'the price sits in the consumer CU's own memory queue'), so fewer resident workgroups per
CU = shorter edges at a smaller prefetch window.  Chain alone, eager, logits checked once per setting.   python tools/decode_chain_residency.py"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def main():
    from aki_amd import ops, _lib
    from aki_amd.phi3 import Phi3ForCausalLM, make_phi3_config
    cfg = make_phi3_config()
    lm = Phi3ForCausalLM(cfg)
    for p in lm.parameters():
        p.data.normal_(0, 0.02)
    lm = lm.to("cuda").to(torch.bfloat16).eval()
    L = 655
    x = torch.randn(1, L, cfg.hidden_size, device="cuda", dtype=torch.bfloat16) * 0.5
    table = ops.MaskTable.from_host([[(4, 148, 4, 148)]], torch.ones(1, L, dtype=torch.bool).numpy(), [L], "cuda")

    def T(f, n=24):
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(n):
            f()
        torch.cuda.synchronize()
        return round((time.perf_counter() - t) * 1e3 / n, 3)

    with torch.no_grad():
        lm.model.use_decode_chain = False
        out = lm(inputs_embeds=x, attention_mask=table, use_cache=True, cache_capacity=L + 136)
        c0, ids0 = out.past_key_values, out.logits[:, -1].float().argmax(-1)
        ref = lm.decode_step(input_ids=ids0, past_key_values=c0).clone()
        lm.model.use_decode_chain = True
    with _lib.use_lab(0) as lab, torch.no_grad():
        for preset, pname in ((0, "{2,2,2,2}"), (7, "{1,1,1,1}"), (4, "{4,4,4,4}"), (8, "{8,4,16,4}")):
            for per_cu, pad in ((4, 0), (3, 36 * 1024), (2, 62 * 1024), (1, 120 * 1024)):
                lab.aki_lab_set_chain_nb(preset)
                lab.aki_lab_set_chain_lds(pad)
                r = {"batches": pname, "max_workgroups_per_cu": per_cu}
                for nowait in (0, 1):
                    lab.aki_lab_set_chain(8, 2, 32, nowait)
                    out = lm(inputs_embeds=x, attention_mask=table, use_cache=True, cache_capacity=L + 136)
                    cache = out.past_key_values
                    got = lm.decode_step(input_ids=ids0, past_key_values=cache)
                    if not nowait:
                        r["logits_differing"] = int((got != ref).sum())
                        cache.chain.check()
                    h = lm.get_input_embeddings()(ids0).reshape(1, -1)
                    cos, sin = lm.model.rotary_emb.tables(cache.capacity, h.device, cache.host_len)
                    ch = cache.chain
                    r["nowait_ms" if nowait else "ms"] = T(lambda: ch.step(h, cos, sin, cache.cache_len, cache.valid_bits, cache.capacity))
                print(json.dumps(r), flush=True)
        lab.aki_lab_set_chain(8, 1, 32, 0)
        lab.aki_lab_set_chain_nb(0)
        lab.aki_lab_set_chain_lds(0)


if __name__ == "__main__":
    main()
