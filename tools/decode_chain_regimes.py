"""The one-launch decode step under different launch conditions (lab library): the same kernel has been seen at 1.2 ms and at
3.4 ms per token.  For each preset of batches per workgroup and number of vector copies: (a) the chain alone, eager, same
position; (b) DecodeGraph replay + argmax (the greedy loop), at two cache capacities; (c) DecodeGraph replay with constant ids.
    python tools/decode_chain_regimes.py"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

# batches per workgroup {qkv, o, gate_up, down} / batches requested before the wait
PRESETS = {0: "product", 25: "{2,2,4,2}/{2,2,4,2} (2 workgroups per CU)", 26: "{2,2,4,2}/{2,2,3,2} (2 per CU)", 12: "{2,2,2,2}/{2,2,2,2}", 13: "{2,2,4,2}/{2,2,2,2}", 14: "{2,2,4,2}/{2,2,2,1}", 16: "{2,2,4,4}/{2,2,2,1}", 17: "{2,2,4,2}/{2,2,1,2}", 19: "{2,2,4,2}/{2,2,1,1}",
           21: "{4,2,8,4}/{2,2,2,1}", 22: "{2,2,4,2}/{1,1,1,1}", 23: "{4,4,8,4}/{1,1,1,1}", 12: "{2,2,2,2}/{2,2,2,2}"}
NOWAIT = 1 if '--nowait' in sys.argv else 0     # 1: the bare weight stream of each structure (wrong results, nothing checked)
ALL = {i: f'preset {i}' for i in range(1, 27)}
TOUCH = (0,)    # touch loads of the batches beyond the register slots while the workgroup waits


def main():
    from aki_amd import ops, _lib
    from aki_amd.phi3 import Phi3ForCausalLM, make_phi3_config, DecodeGraph
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

    rows = []
    # reference logits of 6 greedy steps on the five-launch path: every timed configuration is also CHECKED against them under
    # graph replay (a replayed chain whose counters were not re-zeroed once ran at the bare-stream speed with wrong logits)
    with torch.no_grad():
        lm.model.use_decode_chain = False
        out = lm(inputs_embeds=x, attention_mask=table, use_cache=True, cache_capacity=L + 136)
        c0, ids = out.past_key_values, out.logits[:, -1].float().argmax(-1)
        ref = []
        for _ in range(6):
            l_ = lm.decode_step(input_ids=ids, past_key_values=c0)
            ref.append(l_.clone())
            ids = l_.float().argmax(-1)
        lm.model.use_decode_chain = True
    with _lib.use_lab(0) as lab, torch.no_grad():
        for preset, pname in ((ALL if NOWAIT else PRESETS).items()):
            for touch in ((0,) if NOWAIT else TOUCH if preset not in (0, 12) else (0,)):
                for copies, nowait in ((1, NOWAIT),):
                    lab.aki_lab_set_chain(8, copies, 32, nowait)
                    lab.aki_lab_set_chain_nb(preset)
                    lab.aki_lab_set_chain_touch(touch if preset else -1)
                    r = {"preset": pname, "touch": touch, "copies": copies, "nowait": nowait}
                    for cap in (L + 136,):
                        out = lm(inputs_embeds=x, attention_mask=table, use_cache=True, cache_capacity=cap)
                        cache = out.past_key_values
                        ids = out.logits[:, -1].float().argmax(-1)
                        lm.decode_step(input_ids=ids, past_key_values=cache)
                        h = lm.get_input_embeddings()(ids).reshape(1, -1)
                        cos, sin = lm.model.rotary_emb.tables(cache.capacity, h.device, cache.host_len)
                        ch = cache.chain
                        r[f"eager_chain_only_cap{cap}"] = T(lambda: ch.step(h, cos, sin, cache.cache_len, cache.valid_bits, cache.capacity))
                        st = DecodeGraph(lm, cache)
                        if cap == L + 136 and not nowait:
                            out2 = lm(inputs_embeds=x, attention_mask=table, use_cache=True, cache_capacity=cap)
                            c2, i2 = out2.past_key_values, out2.logits[:, -1].float().argmax(-1)
                            s2 = DecodeGraph(lm, c2)
                            bad = 0
                            for k in range(6):
                                l_ = s2.step(i2)
                                bad += int((l_ != ref[k]).sum())
                                i2 = l_.float().argmax(-1)
                            r["graph_logits_differing_from_five_launch_path"] = bad
                        nxt = [ids]

                        def greedy():
                            nxt[0] = st.step(nxt[0]).float().argmax(-1)
                        greedy()
                        r[f"graph_greedy_cap{cap}"] = T(greedy)
                        r[f"graph_same_ids_cap{cap}"] = T(lambda: st.step(ids))
                        if not nowait:
                            ch.check()
                    rows.append(r)
                    print(json.dumps(r), flush=True)
        lab.aki_lab_set_chain(8, 1, 32, 0)
        lab.aki_lab_set_chain_nb(0)
        lab.aki_lab_set_chain_touch(-1)


if __name__ == "__main__":
    main()
