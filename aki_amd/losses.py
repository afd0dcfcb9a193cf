"""Training-side callers of the model, mirrored from the reference's train/losses.py so that its training loop
(train/train_utils.py:230-252: `loss = loss_fn(model, tokenizer, images, input_ids, attention_mask, autocast)`) runs unchanged on
this stack: the two loss callables (label preparation + `model(...)[0]`) and the cosine learning-rate schedule with warm-up and
a floor.  Host-side glue only - the arithmetic is the model's (lm_head + shifted cross-entropy on the HIP path, §8 a13).

    NextTokenPrediction   train/losses.py:83-116    labels = input ids with padding ignored
    SupervisedPrediction  train/losses.py:119-151   caller-made labels; padding and the model's special tokens ignored
    get_cosine_schedule_with_warmup  train/losses.py:10-40
"""
from __future__ import annotations

import math
from typing import Callable, Optional

import torch
from torch.optim.lr_scheduler import LambdaLR

SUPPORTED_LOSSES = ["next_token_prediction", "supervised_finetune"]
IGNORE_INDEX = -100


def lr_multiplier(step: int, lr: float, min_lr: float, num_warmup_steps: int, num_training_steps: int, num_cycles: float = 0.5) -> float:
    """Factor applied to `lr` at optimizer step `step`.  A plain multiplier m in [0, 1] (linear warm-up, then a cosine) is
    squeezed into [min_lr / lr, 1]: factor = floor + (1 - floor) * m - so the warm-up starts at min_lr, not at zero, and the
    cosine never goes below it."""
    floor = 1.0 - (lr - min_lr) / lr
    if step < num_warmup_steps:
        m = step / max(1, num_warmup_steps)
    else:
        progress = (step - num_warmup_steps) / max(1, num_training_steps - num_warmup_steps)
        m = max(0.0, 0.5 * (1.0 + math.cos(2.0 * math.pi * num_cycles * progress)))
    return floor + (1.0 - floor) * m


def get_cosine_schedule_with_warmup(optimizer, lr, min_lr, num_warmup_steps: int, num_training_steps: int, num_cycles: float = 0.5,
                                    last_epoch: int = -1) -> LambdaLR:
    """Same signature and values as the reference's scheduler factory (a torch LambdaLR over `lr_multiplier`)."""
    return LambdaLR(optimizer, lambda step: lr_multiplier(step, lr, min_lr, num_warmup_steps, num_training_steps, num_cycles), last_epoch)


class TrainerSchedule:
    """The same schedule for `aki_amd.trainer.AkiTrainer` / `AkiShardedTrainer`, whose learning rate is a plain attribute read by
    the fused AdamW kernel: `sched.step()` after every `trainer.optimizer_step()`."""

    def __init__(self, trainer, lr: float, min_lr: float, num_warmup_steps: int, num_training_steps: int, num_cycles: float = 0.5):
        self.trainer, self.args, self.n = trainer, (lr, min_lr, num_warmup_steps, num_training_steps, num_cycles), 0
        self.base_lr = lr
        trainer.lr = lr * lr_multiplier(0, *self.args)

    def step(self) -> float:
        self.n += 1
        self.trainer.lr = self.base_lr * lr_multiplier(self.n, *self.args)
        return self.trainer.lr


def unwrap_model(model):
    """The module behind torch's (Distributed)DataParallel wrappers; anything else is returned as is."""
    wrappers = (torch.nn.DataParallel, torch.nn.parallel.DistributedDataParallel)
    return model.module if isinstance(model, wrappers) else model


class Loss:
    name = None

    def __call__(self, model, tokenizer, images, input_ids, attention_mask, autocast: Callable):
        raise NotImplementedError


class NextTokenPrediction(Loss):
    """Pre-training objective: every non-padding token is a target; the language model shifts (HF convention)."""
    name = "next_token_prediction"

    def __call__(self, model, tokenizer, images: torch.Tensor, input_ids: torch.Tensor, attention_mask: torch.Tensor, autocast: Callable):
        labels = torch.where(input_ids == tokenizer.pad_token_id, torch.full_like(input_ids, IGNORE_INDEX), input_ids)
        with autocast():
            return model(vision_x=images, lang_x=input_ids, attention_mask=attention_mask, labels=labels)[0]


class SupervisedPrediction(Loss):
    """Instruction tuning: the collator supplies the labels (prompt positions already ignored); padding and the model's own special
    tokens (<image>, <|endofchunk|>) are ignored as well.  Like the reference, the caller's `labels` tensor is edited in place."""
    name = "supervised_finetune"

    def __call__(self, model, tokenizer, images: torch.Tensor, input_ids: torch.Tensor, labels: torch.Tensor, attention_mask: torch.Tensor,
                 autocast: Callable, image_size: Optional[torch.Tensor] = None):
        special = torch.as_tensor(list(unwrap_model(model).special_token_ids), device=labels.device, dtype=labels.dtype)
        labels[(labels == tokenizer.pad_token_id) | torch.isin(labels, special)] = IGNORE_INDEX
        with autocast():
            return model(vision_x=images, image_size=image_size, lang_x=input_ids, attention_mask=attention_mask, labels=labels)[0]


def get_loss_fn(loss_name: str) -> Loss:
    for cls in (NextTokenPrediction, SupervisedPrediction):
        if cls.name == loss_name:
            return cls()
    raise ValueError(f"Loss {loss_name} not supported. Supported losses: {SUPPORTED_LOSSES}")
