"""SFT collate on the device: the host-side mirror of train/sft_data_utils/loader_utils.py (`batch_collate_pad` :53-91,
`batchify` :94-121; `_pad_trunc` :11-50 is the kernel) with the same arguments and return structure.

The reference pads and truncates Python lists sample by sample in DataLoader workers and ships three [B, T] LongTensors
plus the stacked images to the GPU afterwards.  Here the ragged token lists are packed once (one pinned staging tensor, one
H2D copy) and `aki_sft_collate_pad` writes the padded [B, T] id / label / mask arrays where the model reads them; images go
into one preallocated device tensor (absent images stay black, as in the reference).  Integer work: bit-exact.
"""
from __future__ import annotations

from typing import List, Optional

import torch

from . import _lib as L
from .ops import _dev, _ptr, _stream

IGNORE_INDEX = -100       # train/sft_data_utils/templates/templates.py
_INFINITE = int(1e12)     # loader_utils.py:7: "no truncation"
_IMG_SIZE = 384           # loader_utils.py:8


def _flat(seqs) -> List[int]:
    out: List[int] = []
    for s in seqs:
        out.extend(s.tolist() if torch.is_tensor(s) else s)
    return out


def batch_collate_pad(batch: list, padding: str, padding_side: str, pad_token_id: int, max_length: Optional[int],
                      device="cuda") -> dict:
    """[{input_ids, labels, attention_mask}, ...] -> {"input_ids", "labels", "attention_mask"}: LongTensors [B, T] on `device`.
    padding "longest": T = the longest sample - `_pad_trunc` (loader_utils.py:30-31) replaces the caller's limit by it, so nothing is
    truncated in this mode; "max_length": T = max_length + 1 (the + 1 is the BOS token, loader_utils.py:78) and longer samples keep
    their first T tokens."""
    assert padding in ["longest", "max_length"]
    assert padding_side in ["left", "right"]
    if padding == "max_length":
        assert max_length is not None, "max_length should be given if padding == 'max_length'"
    else:
        max_length = max_length or _INFINITE
    dev = torch.device(device)
    if dev.type != "cuda":
        raise L.AkiError("sft_collate runs on the MI355X (device='cuda'); there is no CPU fallback")
    lengths = [len(s["input_ids"]) for s in batch]
    for s, n in zip(batch, lengths):
        assert len(s["labels"]) == n and len(s["attention_mask"]) == n, "input_ids / labels / attention_mask of a sample differ in length"
    B = len(batch)
    T = max_length + 1 if padding == "max_length" else max(lengths)
    offs = [0]
    for n in lengths:
        offs.append(offs[-1] + n)
    total = max(offs[-1], 1)
    # one staging tensor, one H2D copy: [3, total] tokens followed by the B + 1 offsets
    stage = torch.empty((3 * total + B + 1,), dtype=torch.int64).pin_memory()
    for i, key in enumerate(("input_ids", "labels", "attention_mask")):
        stage[i * total: i * total + offs[-1]] = torch.as_tensor(_flat(s[key] for s in batch), dtype=torch.int64)
    stage[3 * total:] = torch.as_tensor(offs, dtype=torch.int64)
    d = stage.to(dev, non_blocking=True)
    offsets = d[3 * total:].to(torch.int32)
    out = torch.empty((3, B, T), dtype=torch.int64, device=dev)
    L.check(L.load().aki_sft_collate_pad(_ptr(d), d.data_ptr() + 8 * total, d.data_ptr() + 16 * total, _ptr(offsets), B, T, int(pad_token_id),
                                         IGNORE_INDEX, 1 if padding_side == "left" else 0, _ptr(out[0]), _ptr(out[1]), _ptr(out[2]),
                                         _stream()), "aki_sft_collate_pad")
    return {"input_ids": out[0], "labels": out[1], "attention_mask": out[2]}


def batchify(batch, tokenizer, max_length: int, use_trunc=False, device="cuda"):
    """collate_fn of the SFT DataLoader (loader_utils.py:94-121): (images [B, 3, 384, 384], text batch); samples without an image
    get an all-zero (black) one."""
    text = batch_collate_pad([d["text"] for d in batch], padding="longest" if use_trunc else "max_length", padding_side="right",
                             max_length=max_length, pad_token_id=tokenizer.pad_token_id, device=device)
    images = torch.zeros((len(batch), 3, _IMG_SIZE, _IMG_SIZE), dtype=torch.float32, device=device)
    for i, d in enumerate(batch):
        if d["image"] is not None:
            images[i].copy_(d["image"].reshape(3, _IMG_SIZE, _IMG_SIZE), non_blocking=True)
    return images, text
