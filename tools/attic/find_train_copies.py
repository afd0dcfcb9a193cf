"""Which torch (non-HIP) ops run inside one training step, and from where?  2 decoder layers at full width."""
import os, sys, collections, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from torch.utils._python_dispatch import TorchDispatchMode
from aki_amd.factory import build_aki
from aki_amd.phi3 import make_phi3_config
from aki_amd.siglip import make_siglip_config
from aki_amd.trainer import AkiTrainer
dev = torch.device("cuda", 0)
model = build_aki(make_phi3_config(num_hidden_layers=2), make_siglip_config(num_hidden_layers=2), dtype=torch.bfloat16, device=dev)
model.train(); model.set_trainable()
tr = AkiTrainer(model)
vx, ids, am = bench.synth_batch(8, dev, torch.bfloat16, model.media_token_id, seed=1)
labels = ids.clone(); labels[labels == model.media_token_id] = -100
cnt = collections.Counter()
SKIP = ("view", "reshape", "_unsafe_view", "as_strided", "detach", "alias", "t", "transpose", "permute", "expand", "slice", "select",
        "unsqueeze", "squeeze", "empty", "empty_like", "empty_strided", "_local_scalar_dense", "is_same_size", "sym_size", "split", "chunk",
        "unbind", "lift_fresh", "new_empty", "stride", "sym_stride", "sym_numel", "is_contiguous", "size", "numel", "dim", "storage_offset",
        "sym_storage_offset", "new_empty_strided", "_reshape_alias", "unfold", "narrow", "split_with_sizes", "view_as")


class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.__name__.split(".")[0]
        if name not in SKIP:
            site = "?"
            for fr in reversed(traceback.extract_stack()):
                if "/aki_amd/" in fr.filename:
                    site = f"{os.path.basename(fr.filename)}:{fr.lineno} {fr.line[:60]}"
                    break
            numel = max([a.numel() for a in args if isinstance(a, torch.Tensor)] + [0])
            cnt[(name, site)] += 1
            cnt[("~bytes", name, site)] += numel
        return func(*args, **(kwargs or {}))


tr.train_step(vx, ids, attention_mask=am, labels=labels)
with Spy():
    tr.train_step(vx, ids, attention_mask=am, labels=labels)
rows = [(c, k) for k, c in cnt.items() if k[0] != "~bytes"]
for c, (n, s) in sorted(rows, key=lambda r: -cnt[("~bytes", r[1][0], r[1][1])])[:32]:
    print(f"{c:4d} x {n:22s} {cnt[('~bytes', n, s)]/1e6:9.1f} Melem  {s}")
