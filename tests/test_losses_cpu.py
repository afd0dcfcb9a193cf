"""aki_amd/losses.py against fixtures produced by the reference's own train/losses.py (tests/golden/make_golden.py::g_train_losses):
the LR-schedule multipliers and the labels its two loss callables hand to the model.  CPU only - a stub model records the call."""
import contextlib

import numpy as np
import pytest
import torch

from conftest import load_golden
from golden import gen
from aki_amd import losses as LS


class _Stub:
    def __init__(self, specials):
        self.special_token_ids = specials
        self.seen = {}

    def __call__(self, **kw):
        self.seen = kw
        return (torch.tensor(1.5),)


def test_lr_schedule_equals_the_reference_multipliers():
    g = load_golden("train_losses.npz")
    scheds, steps, *_ = gen.loss_cases()
    for i, sc in enumerate(scheds):
        got = np.array([LS.lr_multiplier(st, **sc) for st in steps])
        np.testing.assert_allclose(got, g[f"mult_{i}"], rtol=1e-12, atol=1e-15)
        opt = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=sc["lr"])
        sched = LS.get_cosine_schedule_with_warmup(opt, **sc)
        lrs = []
        for _ in range(12):
            lrs.append(opt.param_groups[0]["lr"])
            opt.step()
            sched.step()
        np.testing.assert_allclose(lrs, [sc["lr"] * LS.lr_multiplier(s, **sc) for s in range(12)], rtol=1e-12)


def test_trainer_schedule_drives_a_plain_lr_attribute():
    tr = type("T", (), {"lr": 0.0})()
    sc = dict(lr=1e-4, min_lr=1e-5, num_warmup_steps=3, num_training_steps=10)
    s = LS.TrainerSchedule(tr, **sc)
    seq = [tr.lr] + [s.step() for _ in range(5)]
    np.testing.assert_allclose(seq, [sc["lr"] * LS.lr_multiplier(i, **sc) for i in range(6)], rtol=1e-12)


def test_loss_callables_prepare_labels_like_the_reference():
    g = load_golden("train_losses.npz")
    _, _, pad, specials, ids, labels = gen.loss_cases()
    tok = type("Tok", (), {"pad_token_id": pad})()
    m = _Stub(specials)
    ids_t = torch.from_numpy(ids.copy())
    out = LS.get_loss_fn("next_token_prediction")(m, tok, None, ids_t, torch.ones_like(ids_t), contextlib.nullcontext)
    assert float(out) == 1.5 and set(m.seen) == {"vision_x", "lang_x", "attention_mask", "labels"}
    assert np.array_equal(m.seen["labels"].numpy(), g["ntp_labels"]) and np.array_equal(ids_t.numpy(), ids)      # ids untouched
    lab_t = torch.from_numpy(labels.copy())
    wrapped = torch.nn.DataParallel(torch.nn.Identity())
    assert LS.unwrap_model(wrapped) is wrapped.module and LS.unwrap_model(m) is m
    out = LS.get_loss_fn("supervised_finetune")(m, tok, None, ids_t, lab_t, torch.ones_like(ids_t), contextlib.nullcontext)
    assert float(out) == 1.5 and "image_size" in m.seen
    assert np.array_equal(m.seen["labels"].numpy(), g["sft_labels"]) and m.seen["labels"] is lab_t              # edited in place, like the reference
    with pytest.raises(ValueError):
        LS.get_loss_fn("contrastive")
    assert LS.SUPPORTED_LOSSES == ["next_token_prediction", "supervised_finetune"]
