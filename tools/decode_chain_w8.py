"""e4m3-weight decode, batch 1: the one-launch chain under the lab library's batch / prefetch presets against the five-launch path, ONE
process on one box, hipGraph replay + argmax, logits of 6 greedy steps checked against the five-launch path.
    python tools/decode_chain_w8.py"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

PRESETS = {0: "{1,1,1,1}/{1,1,1,1} (product)", 7: "{2,2,2,2}/{1,1,1,1}", 12: "{2,2,2,2}/{2,2,2,2}", 20: "{1,1,2,1}/{1,1,2,1}", 21: "{1,1,1,2}/{1,1,1,2}", 9: "{2,1,4,1}/{2,1,4,1}",
           13: "{2,2,4,2}/{2,2,2,2}", 14: "{2,2,4,2}/{2,2,2,1}"}


def main():
    from aki_amd import ops, _lib
    from aki_amd.phi3 import Phi3ForCausalLM, make_phi3_config, DecodeGraph
    cfg = make_phi3_config()
    lm = Phi3ForCausalLM(cfg)
    for p in lm.parameters():
        p.data.normal_(0, 0.02)
    lm = lm.to("cuda").to(torch.bfloat16).eval()
    lm.enable_fp8()
    L = 655
    x = torch.randn(1, L, cfg.hidden_size, device="cuda", dtype=torch.bfloat16) * 0.5
    table = ops.MaskTable.from_host([[(4, 148, 4, 148)]], torch.ones(1, L, dtype=torch.bool).numpy(), [L], "cuda")

    def run(n_check=6, n_time=48):
        out = lm(inputs_embeds=x, attention_mask=table, use_cache=True, cache_capacity=L + 136)
        cache, ids = out.past_key_values, out.logits[:, -1].float().argmax(-1)
        g = DecodeGraph(lm, cache)
        logits = []
        for _ in range(n_check):
            l_ = g.step(ids)
            logits.append(l_.clone())
            ids = l_.float().argmax(-1)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(n_time):
            ids = g.step(ids).float().argmax(-1)
        torch.cuda.synchronize()
        return logits, round((time.perf_counter() - t) * 1e3 / n_time, 4), cache

    with _lib.use_lab(0) as lab, torch.no_grad():
        lm.model.decode_chain_w8 = False
        ref, ms, _ = run()
        print(json.dumps({"path": "five launches per layer", "ms_per_token": ms}), flush=True)
        lm.model.decode_chain_w8 = True
        for preset, name in PRESETS.items():
            lab.aki_lab_set_chain_nb(preset)
            got, ms, cache = run()
            cache.chain.check()
            bad = sum(int((a != b).sum()) for a, b in zip(got, ref))
            print(json.dumps({"path": "chain " + name, "ms_per_token": ms, "logits_differing_from_five_launch_path": bad}), flush=True)
        lab.aki_lab_set_chain_nb(0)


if __name__ == "__main__":
    main()
