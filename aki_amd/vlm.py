"""VLM base classes: host-side mirror of src/vlm.py (same class / method / attribute names).

Reference: /root/reference/codes/open_flamingo/src/vlm.py
  VLM.__init__ :26-103 | _encode_vision_x :184-207 | VLMWithLanguageStream :379-408
  _make_modality_mutual_mask :410-443 | _prepare_inputs_for_forward :445-603

What changes on MI355X: ``_prepare_inputs_for_forward`` runs the splice kernel (DecoupledEmbedding gather +
vision-token splice + padding) and returns the mask as an ``ops.MaskTable`` - a per-sample rectangle,
valid-column bits and sequence lengths - instead of the dense (B,1,L,L) int64 tensor.  The dense tensor is
still available bit-exactly through ``_make_modality_mutual_mask`` / ``MaskTable`` -> ``ops.mask_dense``.
"""
from __future__ import annotations

from typing import List, Optional, Tuple, Union

import torch
from torch import nn

from . import ops
from .helpers import DecoupledEmbedding, DecoupledLinear, VLMOutputWithPast
from .utils import getattr_recursive, num_params

ASSISTANT_TOKEN_ID = 32001  # hard-coded by the reference, src/vlm.py:490-496


def _activate_checkpointing(module: nn.Module) -> None:
    """Route `module.forward` through torch.utils.checkpoint while gradients are being recorded (inference and no-grad calls go straight
    through, as do calls with a KV cache - a prefill keeps nothing for backward anyway)."""
    import torch.utils.checkpoint as ckpt
    inner = module.forward

    def forward(*args, **kwargs):
        recording = torch.is_grad_enabled() and (any(isinstance(a, torch.Tensor) and a.requires_grad for a in args)
                                                 or any(p.requires_grad for p in module.parameters()))
        if recording:
            return ckpt.checkpoint(inner, *args, use_reentrant=False, **kwargs)
        return inner(*args, **kwargs)

    module.forward = forward
    module._ckpt_active = True


class VLM(nn.Module):
    """Vision tower + vision tokenizer + language model under the attribute names checkpoints and the training scripts rely
    on (`vision_encoder`, `vision_tokenizer`, `lang_model`; src/vlm.py:26-103).  Construction swaps the language model's
    token tables for their decoupled forms so that the special tokens get trainable rows of their own."""

    def __init__(self, vision_encoder: nn.Module, vision_tokenizer: nn.Module, lang_model: nn.Module,
                 initial_tokenizer_len: int, pad_token_id: int, gradient_checkpointing: bool = False,
                 base_img_size: Optional[int] = None):
        super().__init__()
        self.vision_encoder, self.vision_tokenizer, self.lang_model = vision_encoder, vision_tokenizer, lang_model
        cfg = lang_model.config
        self.lang_embedding_dim = lang_model.get_input_embeddings().weight.shape[1]
        self.lang_hidden_dim = getattr(cfg, "d_model", None) or cfg.hidden_size
        self.vis_embedding_dim = vision_tokenizer.dim_media
        self.num_tokens_per_vis = vision_tokenizer.num_tokens_per_media
        if base_img_size is None:       # native resolution of the tower: HF towers carry it in their config, open_clip as a tuple
            vcfg = getattr(vision_encoder, "config", None)
            base_img_size = vcfg.image_size if vcfg is not None else vision_encoder.image_size[0]
        self.base_img_size = base_img_size
        self.pad_token_id = pad_token_id
        self.initial_tokenizer_len = initial_tokenizer_len
        self._decouple_token_tables(n_new=len(self.special_tokens), std=getattr(cfg, "initializer_range", 0.02))
        # As in the reference (src/vlm.py:63, :124-127) the flag only MARKS modules; `init_gradient_checkpointing()` (below; the reference's
        # driver calls its twin at train/train.py:315-327, AkiTrainer calls this one) switches recomputation on for the marked ones.
        self._gradient_checkpointing = bool(gradient_checkpointing)
        self.vision_tokenizer._use_gradient_checkpointing = gradient_checkpointing

    def init_gradient_checkpointing(self):
        """Activation checkpointing for every module marked `_use_gradient_checkpointing` (src/vlm.py:360-378: non-reentrant
        `checkpoint_wrapper` around the decoder blocks and the vision tokenizer).  Same effect here without wrapper modules - parameter
        names and state-dict keys stay as they are: a marked module's training forward runs under `torch.utils.checkpoint` (non-reentrant),
        so only its input is kept and its HIP forward kernels run again inside backward.  The kernels are deterministic (no float atomics), so
        the recomputed activations - and therefore the gradients - are bit-identical to a run without checkpointing
        (tests/test_train_gpu.py::test_gradient_checkpointing_recomputes_and_changes_nothing)."""
        n = 0
        for m in self.modules():
            if getattr(m, "_use_gradient_checkpointing", False) and not getattr(m, "_ckpt_active", False):
                _activate_checkpointing(m)
                n += 1
        return n

    def _decouple_token_tables(self, n_new: int, std: float) -> None:
        """Input embedding -> DecoupledEmbedding, output head -> DecoupledLinear; the original rows are shared with the
        incoming modules, the new rows start as N(0, std) (state-dict keys: `...embed_tokens.additional_embedding.weight`,
        `lm_head.additional_fc.{weight,bias}`; like the reference the head's `bias=True` default stays in force even for
        a bias-free lm_head, so checkpoints carry `lm_head.bias`)."""
        last_original = self.initial_tokenizer_len - 1
        emb_in = self.lang_model.get_input_embeddings()
        table = DecoupledEmbedding(max_original_id=last_original, num_additional_embeddings=n_new, _weight=emb_in.weight,
                                   pad_token_id=self.pad_token_id)
        head_in = self.lang_model.get_output_embeddings()
        head = DecoupledLinear(max_original_id=last_original, additional_out_features=n_new, _weight=head_in.weight,
                               _bias=getattr(head_in, "bias", None))
        for new_mod, ref in ((table, emb_in.weight), (head, head_in.weight)):
            new_mod.to(device=ref.device, dtype=ref.dtype)
        with torch.no_grad():
            if n_new > 0:
                table.additional_embedding.weight.normal_(mean=0.0, std=std)
                head.additional_fc.weight.normal_(mean=0.0, std=std)
        self.lang_model.set_input_embeddings(table)
        self.lang_model.set_output_embeddings(head)

    # ---- vision ------------------------------------------------------------------------------------------
    def _encode_vision_x(self, vision_x: torch.Tensor):
        """(b, T_img, F, C, H, W) -> (b, T_img, F, v, d)   (src/vlm.py:184-207)."""
        assert vision_x.ndim == 6, "vision_x should be of shape (b, T_img, F, C, H, W)"
        b, T, Fr = vision_x.shape[:3]
        x = vision_x.reshape(b * T * Fr, *vision_x.shape[3:])
        with torch.no_grad():
            interp = x.shape[-1] != self.base_img_size
            x = self.vision_encoder(x, interpolate_pos_encoding=interp).last_hidden_state
        return x.reshape(b, T, Fr, x.shape[1], x.shape[2])

    # ---- what the training scripts ask the model -------------------------------------------------------------
    def set_trainable(self):
        raise NotImplementedError

    def _should_apply_weight_decay(self, parameter_name):
        raise NotImplementedError

    def group_params_by_weight_decay(self):
        """(decayed, not decayed) trainable parameters, in registration order (train/train.py:330-337 builds AdamW from it)."""
        decayed, plain = [], []
        for name, p in self.named_parameters():
            if p.requires_grad:
                (decayed if self._should_apply_weight_decay(name) else plain).append(p)
        return decayed, plain

    @property
    def num_trainable_params(self):
        return num_params(self, filter_to_trainable=True)

    def _param_report(self, trainable_only: bool) -> str:
        what = "trainable parameters" if trainable_only else "parameters"
        parts = (("Vision encoder", self.vision_encoder), ("Vision tokenizer", self.vision_tokenizer), ("Language model", self.lang_model))
        return "\n".join(f"{label}: {num_params(mod, filter_to_trainable=trainable_only):,} {what}" for label, mod in parts)

    @property
    def num_params_per_module(self):
        return self._param_report(False)

    @property
    def num_trainable_params_per_module(self):
        return self._param_report(True)

    # ---- special tokens (<image>, <|endofchunk|>) ------------------------------------------------------------------
    @property
    def special_tokens(self):
        """{attribute stem: token string}; subclasses fill `_special_tokens` BEFORE calling this constructor."""
        assert "media_token" in self._special_tokens, "a VLM needs a media token: define _special_tokens['media_token'], " \
            "add it to the tokenizer and hand its id to set_special_token_ids"
        return self._special_tokens

    @property
    def special_token_ids(self):
        return [getattr(self, f"{stem}_id") for stem in self.special_tokens]

    def set_special_token_ids(self, string_to_ids):
        """{token string: id} from the tokenizer -> `self.<stem>_id`, mirrored onto the language model."""
        missing = [tok for tok in self.special_tokens.values() if tok not in string_to_ids]
        assert not missing, f"no id given for the special tokens {missing}"
        for stem, tok in self.special_tokens.items():
            for holder in (self, self.lang_model):
                setattr(holder, f"{stem}_id", string_to_ids[tok])


class VLMWithLanguageStream(VLM):
    """VLM that fuses modalities by inserting vision tokens directly into the language stream."""

    def __init__(self, vision_encoder, vision_tokenizer, lang_model, initial_tokenizer_len, pad_token_id,
                 decoder_layers_attr_name=None, gradient_checkpointing=False, base_img_size=None):
        super().__init__(vision_encoder=vision_encoder, vision_tokenizer=vision_tokenizer, lang_model=lang_model,
                         initial_tokenizer_len=initial_tokenizer_len, pad_token_id=pad_token_id,
                         base_img_size=base_img_size, gradient_checkpointing=gradient_checkpointing)
        self.decoder_layers_attr_name = decoder_layers_attr_name
        for block in getattr_recursive(self.lang_model, self.decoder_layers_attr_name):
            block._use_gradient_checkpointing = gradient_checkpointing
        self.allow_multi_image = False  # the reference cannot splice a second image

    @staticmethod
    def _make_modality_mutual_mask(attention_mask_2d: torch.Tensor, image_start_idx: int, text_start_idx: int,
                                   text_end_idx: int, input_ids_shape: torch.Size, dtype: torch.dtype,
                                   device: torch.device):
        """Dense (1,n,n) int64 0/1 mask, bit-exact with src/vlm.py:410-443, built by the HIP mask kernel from the
        same rectangle the attention kernel consumes (kept for API compatibility and for parity tests)."""
        n = int(input_ids_shape[0])
        rs, re_, _ = slice(int(image_start_idx), int(text_start_idx)).indices(n)
        cs, ce, _ = slice(int(text_start_idx), int(text_end_idx)).indices(n)
        rect = (rs, re_, cs, ce) if (re_ > rs and ce > cs) else (0, 0, 0, 0)
        table = ops.MaskTable.from_host([[rect]], attention_mask_2d.detach().cpu().numpy()[None], None, device)
        return ops.mask_dense(table, 1)[0].to(dtype)

    def _prepare_inputs_for_forward(self, vision_tokens: torch.Tensor, lang_x: torch.Tensor, attention_mask: torch.Tensor,
                                    labels: torch.Tensor = None, past_key_values=None,
                                    vision_attention_mask: Optional[torch.Tensor] = None,
                                    past_media_locations: torch.Tensor = None, past_vision_tokens: torch.Tensor = None,
                                    padding_side: str = "left", num_beams: int = 1, splice_plan=None):
        """src/vlm.py:445-603.  ``attention_mask`` in the returned dict is an ``ops.MaskTable``.  ``splice_plan``: the result of
        `_start_splice_plan(lang_x)` issued before the vision side (saves the stream drain that sizing the outputs costs)."""
        if past_key_values is not None:
            # src/vlm.py:463-468: the caller's mask must span the cached tokens (incl. image tokens) + the new ids
            past_len = past_key_values.get_seq_length()
            assert attention_mask is None or attention_mask.shape[1] == past_len + lang_x.shape[1], (
                "Attention_mask must be as long as the entire past len (including image tokens) and current input IDs. "
                "Check that you've expanded the attention mask to account for past image tokens.")
            if vision_tokens is not None:
                raise NotImplementedError("new images on top of an existing KV cache: the reference cannot do it either "
                                          "(its mask for the new chunk would not cover the cached columns); start a new prefill")
            return {"input_ids": lang_x, "attention_mask": None, "labels": labels}
        if vision_tokens is None:
            return {"input_ids": lang_x, "attention_mask": attention_mask, "labels": labels}
        emb = self.lang_model.get_input_embeddings()
        if attention_mask is None:
            attention_mask = torch.ones_like(lang_x)
        max_rects = ops.L.AKI_MAX_RECTS if self.allow_multi_image else 1
        try:
            embeds, new_labels, table, plan = ops.splice(
                lang_x, attention_mask, labels, emb.weight, emb.additional_embedding.weight, emb.max_original_id,
                vision_tokens, self.media_token_id, self.pad_token_id, ASSISTANT_TOKEN_ID, padding_side, max_rects,
                plan=splice_plan)
        except ops.AkiError as e:
            if "max_rects" in str(e):
                raise RuntimeError("Tensors must have same number of dimensions: got 3 and 1 - the reference cannot "
                                   "splice a second image into one sample (src/vlm.py:547-554); set "
                                   "model.allow_multi_image = True for the build-defined multi-image mask") from e
            raise
        if torch.is_grad_enabled() and embeds.dtype == torch.bfloat16 and (
                vision_tokens.requires_grad or emb.weight.requires_grad or emb.additional_embedding.weight.requires_grad):
            from . import train_ops as T    # training: attach the backward of the splice to the kernel's output
            pos_lang, pos_vis = T.splice_positions(plan, tuple(lang_x.shape), vision_tokens.shape[1], vision_tokens.shape[2],
                                                   embeds.shape[1], padding_side, embeds.device)
            embeds = T.SpliceGradFn.apply(embeds, vision_tokens, emb.weight, emb.additional_embedding.weight,
                                          lang_x.to(torch.int64), pos_lang, pos_vis, emb.max_original_id)
        return {"inputs_embeds": embeds, "attention_mask": table, "labels": new_labels}

    def _start_splice_plan(self, lang_x: torch.Tensor):
        """Issue the splice's planning kernel and its device->host copy now (see ops.splice_plan_async); None off the GPU path."""
        if not lang_x.is_cuda:
            return None
        return ops.splice_plan_async(lang_x, self.media_token_id, ASSISTANT_TOKEN_ID, self.num_tokens_per_vis)

    def _post_forward_hook(self):
        pass

    def get_fsdp_lambda_fn(self):
        """Predicate `module -> bool` naming the sharding units of the model - every decoder block and the vision tokenizer
        (the granularity of train/distributed.py:170-222).  `aki_amd.trainer.AkiShardedTrainer` cuts its gather /
        reduce-scatter buckets at exactly these module boundaries."""
        blocks = getattr_recursive(self.lang_model, self.decoder_layers_attr_name)
        block_ids = {id(b) for b in blocks}
        return lambda module: module is self.vision_tokenizer or id(module) in block_ids

    def group_params_by_weight_decay(self):
        """As the base class, except that the token embedding tables are never decayed (src/vlm.py:690-703)."""
        decayed, plain = [], []
        for name, p in self.named_parameters():
            if p.requires_grad:
                (plain if "lang_model.model.embed_tokens" in name else decayed).append(p)
        return decayed, plain
