"""Where do the copy / fill kernels of one forward come from?  TorchDispatchMode intercepts the aten ops and records the
innermost aki_amd frame that issued them.    python tools/find_copies.py"""
import os, sys, collections, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from torch.utils._python_dispatch import TorchDispatchMode
from aki_amd.factory import build_aki
from aki_amd.phi3 import make_phi3_config
from aki_amd.siglip import make_siglip_config
dev = torch.device("cuda", 0)
model = build_aki(make_phi3_config(num_hidden_layers=2), make_siglip_config(num_hidden_layers=2), dtype=torch.bfloat16, device=dev)
model.eval()
vx, ids, am = bench.synth_batch(8, dev, torch.bfloat16, model.media_token_id, seed=1)
cnt = collections.Counter()
WATCH = ("copy_", "clone", "_to_copy", "fill_", "zero_", "cat", "zeros", "index", "where", "embedding", "add", "mul")


class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.__name__.split(".")[0]
        if name in WATCH:
            site = "?"
            for fr in reversed(traceback.extract_stack()):
                if "/aki_amd/" in fr.filename:
                    site = f"{os.path.basename(fr.filename)}:{fr.lineno} {fr.line[:70]}"
                    break
            numel = max([a.numel() for a in args if isinstance(a, torch.Tensor)] + [0])
            cnt[(name, site, numel)] += 1
        return func(*args, **(kwargs or {}))


with torch.no_grad():
    model(vx, ids, attention_mask=am)
    with Spy():
        model(vx, ids, attention_mask=am)
for (n, s, ne), c in cnt.most_common(30):
    print(c, n, ne, s)
