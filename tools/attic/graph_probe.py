"""Does hipGraph replay shrink the inter-kernel gaps of the (static-shape) forward?  eager vs graph, full AKI-4B."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from aki_amd.factory import build_aki
dev = torch.device("cuda", 0)
model = build_aki(dtype=torch.bfloat16, device=dev).eval()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
vx, ids, am = bench.synth_batch(B, dev, torch.bfloat16, model.media_token_id, seed=1)


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


with torch.no_grad():
    feats = model._encode_vision_x(vx)
    vt = model.vision_tokenizer(feats)
    prep = model._prepare_inputs_for_forward(vision_tokens=vt, lang_x=ids, attention_mask=am, padding_side="right")
    emb, table = prep["inputs_embeds"], prep["attention_mask"]
    lm = lambda: model.lang_model(inputs_embeds=emb, attention_mask=table).logits
    vis = lambda: model.vision_tokenizer(model._encode_vision_x(vx))
    print("eager  LM stack %.3f ms   vision+connector %.3f ms   full forward %.3f ms" % (timeit(lm), timeit(vis), timeit(lambda: model(vx, ids, attention_mask=am))))
    outs = {}
    for name, fn in (("lm", lm), ("vis", vis)):
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            fn()
        torch.cuda.current_stream().wait_stream(s)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            outs[name] = fn()
        print("graph  %-4s %.3f ms" % (name, timeit(g.replay)))
