"""Where a phase edge of the one-launch decode step spends its time (lab library: wall-clock stamps, 100 MHz, of every workgroup of ONE layer).
Per phase: when its workgroups started, saw their READY flag, had x staged, finished computing and had arrived; per edge: from the last
producer's arrival to the consumers' flag sighting, staging, compute, drain + arrival.     python tools/decode_chain_edges.py [--layer 16]"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layer", type=int, default=16)
    ap.add_argument("--preset", type=int, default=0)
    ap.add_argument("--dump", default=None, help="save the raw stamps [phase][workgroup][slot] (.npy)")
    ap.add_argument("--batch", type=int, default=1, help="2..8: the batched chain")
    ap.add_argument("--nowait", action="store_true", help="no dependency waits (wrong results): the lifetimes of the workgroups of the bare stream")
    ap.add_argument("--fp8", action="store_true", help="e4m3 weights through the chain (the product runs them on five launches per layer)")
    a = ap.parse_args()
    from aki_amd import ops, _lib
    from aki_amd.phi3 import Phi3ForCausalLM, make_phi3_config
    cfg = make_phi3_config()
    lm = Phi3ForCausalLM(cfg)
    for p in lm.parameters():
        p.data.normal_(0, 0.02)
    lm = lm.to("cuda").to(torch.bfloat16).eval()
    if a.fp8:
        lm.enable_fp8()
        lm.model.decode_chain_w8 = True
    L = 655
    B = a.batch
    x = torch.randn(B, L, cfg.hidden_size, device="cuda", dtype=torch.bfloat16) * 0.5
    table = ops.MaskTable.from_host([[(4, 148, 4, 148)]] * B, torch.ones(B, L, dtype=torch.bool).numpy(), [L] * B, "cuda")
    names = ["qkv", "attention", "o_proj", "gate_up", "down"]
    with _lib.use_lab(0) as lab, torch.no_grad():
        lab.aki_lab_set_chain_nb(a.preset)
        from aki_amd import _lib
        _lib._lib = _lib.load_lab()            # the batched chain is compiled into the lab library only (round 6)
        lm.model.use_decode_chain_batched = True
        if a.nowait:
            lab.aki_lab_set_chain(8, 1, 32, 1)
        out = lm(inputs_embeds=x, attention_mask=table, use_cache=True, cache_capacity=L + 136)
        cache = out.past_key_values
        ids = out.logits[:, -1].float().argmax(-1)
        for _ in range(3):
            lm.decode_step(input_ids=ids, past_key_values=cache)
        stamps = torch.zeros((5, 2048, 8), dtype=torch.int64, device="cuda")
        lab.aki_lab_set_chain_stamps(stamps.data_ptr(), a.layer)
        lm.decode_step(input_ids=ids, past_key_values=cache)
        torch.cuda.synchronize()
        lab.aki_lab_set_chain_stamps(None, -1)
        lab.aki_lab_set_chain(8, 1, 32, 0)
        lab.aki_lab_set_chain_nb(0)
    s = stamps.cpu().numpy().astype(np.float64) / 100.0          # microseconds
    if a.dump:
        np.save(a.dump, stamps.cpu().numpy())
    t0 = s[s > 0].min()
    s = np.where(s > 0, s - t0, np.nan)
    rep = {}
    prev_arrive = None
    for ph, nm in enumerate(names):
        st, seen, staged, done, arr = (s[ph, :1024, k] for k in (0, 1, 2, 3, 5))
        if ph == 1:
            arr = s[1, 1024:1024 + 32 * a.batch, 5]  # per-(sequence, head) mergers
        n = int(np.isfinite(st).sum())
        r = {"workgroups": n, "first_start": round(float(np.nanmin(st)), 2), "last_start": round(float(np.nanmax(st)), 2),
             "first_flag_seen": round(float(np.nanmin(seen)), 2), "last_flag_seen": round(float(np.nanmax(seen)), 2)}
        if ph != 1:
            r.update(x_staged_after_flag_med=round(float(np.nanmedian(staged - seen)), 2), compute_med=round(float(np.nanmedian(done - staged)), 2),
                     compute_max=round(float(np.nanmax(done - staged)), 2), drain_and_arrive_med=round(float(np.nanmedian(arr - done)), 2),
                     last_compute_done=round(float(np.nanmax(done)), 2))
        if ph == 1:     # inside the attention items (wave 0 of each workgroup): q rotated, P V done, partial drained + ticket back
            rope, pv, tick = (s[1, :1024, k] for k in (2, 3, 4))
            r.update(q_rotated_after_flag_med=round(float(np.nanmedian(rope - seen)), 2), scores_v_pv_med=round(float(np.nanmedian(pv - rope)), 2),
                     partial_drain_ticket_med=round(float(np.nanmedian(tick - pv)), 2), last_ticket=round(float(np.nanmax(tick)), 2))
        life = s[ph, :1024, 5] - s[ph, :1024, 0]
        if ph == 1:
            life = s[1, :1024, 4] - s[1, :1024, 0]
        r["workgroup_lifetime_med"] = round(float(np.nanmedian(life)), 2)
        r["workgroup_lifetime_p90"] = round(float(np.nanpercentile(life[np.isfinite(life)], 90)), 2)
        r["last_arrival"] = round(float(np.nanmax(arr)), 2)
        if prev_arrive is not None:
            r["edge_last_producer_arrival_to_first_flag_seen"] = round(float(np.nanmin(seen)) - prev_arrive, 2)
            r["edge_to_median_flag_seen"] = round(float(np.nanmedian(seen)) - prev_arrive, 2)
        prev_arrive = float(np.nanmax(arr))
        rep[nm] = r
        print(nm, json.dumps(r))
    print("layer span (first qkv flag seen -> last down arrival): %.2f us" % (rep["down"]["last_arrival"] - rep["qkv"]["first_flag_seen"]))


if __name__ == "__main__":
    main()
