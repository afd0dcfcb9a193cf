#!/usr/bin/env python3
"""The batched decode chain (2..8 sequences) at its ring presets, with and without the dependency waits (nowait: wrong results - the bare
weight stream of the structure), next to the five-launch batched step.  Lab library.   python tools/attic/batched_chain_regimes.py [--batch 8]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--steps", type=int, default=48)
    a = ap.parse_args()
    from aki_amd import ops, _lib
    from aki_amd.phi3 import Phi3ForCausalLM, make_phi3_config
    cfg = make_phi3_config()
    lm = Phi3ForCausalLM(cfg)
    for p in lm.parameters():
        p.data.normal_(0, 0.02)
    lm = lm.to("cuda").to(torch.bfloat16).eval()
    B, L = a.batch, 655
    x = torch.randn(B, L, cfg.hidden_size, device="cuda", dtype=torch.bfloat16) * 0.5
    table = ops.MaskTable.from_host([[(4, 148, 4, 148)]] * B, torch.ones(B, L, dtype=torch.bool).numpy(), [L] * B, "cuda")

    def run(batched):
        lm.model.use_decode_chain_batched = batched
        out = lm(inputs_embeds=x, attention_mask=table, use_cache=True, cache_capacity=L + 2 * a.steps + 8)
        cache = out.past_key_values
        ids = out.logits[:, -1].float().argmax(-1)
        for _ in range(4):
            lm.decode_step(input_ids=ids, past_key_values=cache)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            lm.decode_step(input_ids=ids, past_key_values=cache)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e3 / a.steps

    with _lib.use_lab(0) as lab, torch.no_grad():
        print(f"batch {B}: five launches per layer (eager, host-bound?): {run(False):.3f} ms per step")
        for preset in (0, 1, 2, 3, 4):
            lab.aki_lab_set_chain_nb(preset)
            row = []
            for nowait in (0, 1):
                lab.aki_lab_set_chain(8, 1, 32, nowait)
                row.append(run(True))
            print(f"  preset {preset}: with waits {row[0]:.3f} ms, bare stream {row[1]:.3f} ms", flush=True)
        lab.aki_lab_set_chain(8, 1, 32, 0)
        lab.aki_lab_set_chain_nb(0)


if __name__ == "__main__":
    main()
