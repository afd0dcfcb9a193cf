"""Deterministic synthetic parameters/inputs shared by the golden generator and the tests.

Weights are never stored in fixtures: both sides regenerate them from ``numpy.random.Generator(PCG64(seed))``
by walking the state-dict keys in sorted order.  Only (name, shape) lists and expected outputs are committed.
"""
from __future__ import annotations

import zlib
from typing import Dict, Iterable, List, Sequence, Tuple

import numpy as np


def _scale_for(name: str, shape) -> Tuple[float, float]:
    """(mean, std) per parameter kind: 1-D ``weight`` tensors are norm gains (around 1)."""
    leaf = name.split(".")[-1]
    if leaf == "latents":
        return 0.0, 1.0
    if leaf == "weight" and len(shape) == 1:
        return 1.0, 0.1
    if leaf == "bias":
        return 0.0, 0.02
    if "position_embedding" in name:
        return 0.0, 0.1
    return 0.0, 0.05


def fill_params(shapes: Sequence[Tuple[str, Tuple[int, ...]]], seed: int) -> Dict[str, np.ndarray]:
    """Fill tensors in sorted-name order.  Each tensor gets its own stream keyed by (seed, crc32(name))
    so adding/removing a key does not shift the others."""
    out = {}
    for name, shape in sorted(shapes, key=lambda t: t[0]):
        rng = np.random.Generator(np.random.PCG64([seed, zlib.crc32(name.encode())]))
        mean, std = _scale_for(name, shape)
        out[name] = (mean + std * rng.standard_normal(tuple(shape), dtype=np.float32)).astype(np.float32)
    return out


def rng_for(tag: str, seed: int = 0) -> np.random.Generator:
    return np.random.Generator(np.random.PCG64([seed, zlib.crc32(tag.encode())]))


def pack_mask_bits(mask01: np.ndarray) -> np.ndarray:
    return np.packbits(mask01.astype(np.uint8).reshape(-1))


def unpack_mask_bits(bits: np.ndarray, shape) -> np.ndarray:
    n = int(np.prod(shape))
    return np.unpackbits(bits)[:n].reshape(shape).astype(np.int64)


# Structural constants of the tiny end-to-end model (true head_dim 96 is kept).
TINY = dict(
    lm_hidden=192, lm_heads=2, lm_layers=2, lm_inter=256, vocab=32011,
    vis_hidden=64, vis_heads=2, vis_layers=2, vis_inter=128, image=56, patch=14,
    num_vision_tokens=8, media_token_id=32011, eoc_token_id=32012, pad_token_id=32000,
)


def mask_cases() -> List[Tuple[np.ndarray, int, int, int]]:
    """(1-D attention mask, image_start, text_start, text_end) cases for ``_make_modality_mutual_mask``."""
    cases = []
    ones = lambda n: np.ones(n, dtype=np.int64)
    cases.append((np.array([1] * 10 + [0] * 2), 2, 6, 9))            # SURVEY 3.2 probe
    cases.append((ones(12), 0, 0, 0))                                 # no image, no <|assistant|>
    cases.append((ones(12), 0, 0, 7))                                 # no image, q=7 -> rows empty
    cases.append((ones(20), 3, 11, 8))                                # pre-training prompt: q=0 -> te<ts, empty
    cases.append((ones(20), 3, 11, 11))                               # te == ts
    cases.append((ones(20), 3, 11, 19))
    cases.append((ones(20), 3, 11, 20))
    cases.append((ones(20), 3, 11, 27))                               # te beyond n (clipped by slicing)
    cases.append((ones(20), 0, 8, 15))                                # image at position 0
    cases.append((ones(20), 12, 20, 20))                              # image at the very end
    cases.append((np.array([1] * 15 + [0] * 5), 3, 11, 14))          # right padding after answer
    cases.append((np.array([1] * 12 + [0] * 8), 3, 11, 16))          # padding inside the unlock range
    cases.append((np.array([0] * 4 + [1] * 16), 6, 14, 18))          # left padding
    cases.append((np.array([1, 1, 0, 1, 1, 1, 0, 1, 1, 1, 1, 1, 0, 1, 1, 1]), 2, 10, 14))  # holes
    cases.append((np.zeros(9, dtype=np.int64), 1, 5, 8))             # everything masked
    cases.append((ones(1), 0, 0, 0))
    cases.append((ones(2), 0, 1, 2))
    rng = rng_for("mask_cases")
    for _ in range(12):
        n = int(rng.integers(5, 97))
        nv = int(rng.integers(1, max(2, n // 2)))
        s = int(rng.integers(0, n - nv + 1))
        q = int(rng.integers(0, n + 4))
        am = (rng.random(n) > 0.15).astype(np.int64) if rng.random() < 0.4 else np.concatenate(
            (np.ones(n - (k := int(rng.integers(0, n // 3 + 1))), dtype=np.int64), np.zeros(k, dtype=np.int64)))
        cases.append((am, s, s + nv, q + nv))
    cases.append((ones(207), 6, 150, 190 + 144 - 144))                # config-1-like length
    cases.append((np.concatenate((np.ones(600, dtype=np.int64), np.zeros(55, dtype=np.int64))), 6, 150, 638))  # config-2-like
    return cases


def grad_sample_idx(numel: int, k: int = 48) -> np.ndarray:
    """Deterministic flat indices at which gradient fixtures sample a parameter's gradient."""
    if numel <= k:
        return np.arange(numel, dtype=np.int64)
    return (np.arange(k, dtype=np.int64) * 2654435761 + 12345) % numel


def sft_cases():
    """Seeded ragged batches + collate arguments for the SFT collate (train/sft_data_utils/loader_utils.py)."""
    rng = rng_for("sft_collate")
    cases = []
    for (B, lo, hi, padding, side, max_length) in [(4, 3, 40, "max_length", "right", 24), (5, 1, 30, "longest", "right", 16), (3, 10, 60, "max_length", "left", 63),
                                                    (6, 2, 90, "longest", "left", None), (1, 7, 8, "max_length", "right", 4), (8, 400, 700, "max_length", "right", 512),
                                                    (2, 5, 6, "longest", "right", 2)]:
        batch = []
        for _ in range(B):
            n = int(rng.integers(lo, hi + 1))
            ids = rng.integers(0, 32064, n).astype(np.int64)
            lab = np.where(rng.random(n) < 0.4, -100, ids)
            batch.append({"input_ids": ids.tolist(), "labels": lab.tolist(), "attention_mask": [1] * n})
        cases.append((batch, padding, side, 32000, max_length))
    return cases


def loss_cases():
    """Inputs of the loss-callable / LR-schedule fixtures (tests/golden/train_losses.npz): schedule settings and the steps at which the
    multiplier is sampled; seeded id / label batches with padding and special tokens sprinkled in."""
    scheds = [dict(lr=1e-4, min_lr=1e-5, num_warmup_steps=10, num_training_steps=100, num_cycles=0.5),
              dict(lr=3e-5, min_lr=0.0, num_warmup_steps=0, num_training_steps=37, num_cycles=0.5),
              dict(lr=2e-4, min_lr=2e-5, num_warmup_steps=25, num_training_steps=40, num_cycles=1.0)]
    steps = [0, 1, 2, 5, 9, 10, 11, 17, 25, 36, 37, 39, 40, 50, 99, 100, 150]
    rng = rng_for("loss_cases")
    pad, specials = 7, [11, 12]
    ids = rng.integers(13, 60, size=(3, 12)).astype(np.int64)
    ids[0, 9:] = pad
    ids[1, 4] = specials[0]
    ids[2, [2, 7]] = specials
    labels = ids.copy()
    labels[:, :3] = -100                                   # prompt positions already ignored by the collator
    return scheds, steps, pad, specials, ids, labels
