#!/usr/bin/env python3
"""The layer loops issued by the library (csrc/stack.hip) against the per-layer Python loops, alternating in ONE process: the headline batch
(8 x 655, GPU-bound either way) and the one-sample prefill (where the host was the bound).  ms per forward, median of the rounds.
    python tools/stack_ab.py [--batches 8,1] [--rounds 5] [--iters 10]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", default="8,1")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=10)
    a = ap.parse_args()
    import bench
    from aki_amd.factory import build_aki
    dev = torch.device("cuda", 0)
    model = build_aki(dtype=torch.bfloat16, device=dev, seed=0).eval()

    def set_stack(on):
        model.lang_model.model.use_layer_stack = on
        model.vision_encoder.encoder.use_layer_stack = on

    for B in [int(b) for b in a.batches.split(",")]:
        vx, ids, am = bench.synth_batch(B, dev, torch.bfloat16, model.media_token_id, seed=1000)
        res = {True: [], False: []}
        outs = {}
        with torch.no_grad():
            for on in (True, False):
                set_stack(on)
                for _ in range(3):
                    outs[on] = model(vx, ids, attention_mask=am).logits
            assert torch.equal(outs[True], outs[False]), "the two paths disagree"
            for _ in range(a.rounds):
                for on in (True, False):
                    set_stack(on)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(a.iters):
                        model(vx, ids, attention_mask=am)
                    torch.cuda.synchronize()
                    res[on].append((time.perf_counter() - t0) / a.iters * 1e3)
        set_stack(True)
        med = lambda v: sorted(v)[len(v) // 2]
        print(json.dumps({"batch": B, "L": 655, "ms_per_forward_one_call_loops": round(med(res[True]), 3), "ms_per_forward_python_loops": round(med(res[False]), 3),
                          "all_rounds_one_call": [round(x, 3) for x in res[True]], "all_rounds_python": [round(x, 3) for x in res[False]], "logits_equal": True}), flush=True)


if __name__ == "__main__":
    main()
