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


class VLM(nn.Module):
    def __init__(self, vision_encoder: nn.Module, vision_tokenizer: nn.Module, lang_model: nn.Module,
                 initial_tokenizer_len: int, pad_token_id: int, gradient_checkpointing: bool = False,
                 base_img_size: Optional[int] = None):
        super().__init__()
        self.lang_embedding_dim = lang_model.get_input_embeddings().weight.shape[1]
        self.lang_hidden_dim = getattr(lang_model.config, "d_model", None) or lang_model.config.hidden_size
        self.vis_embedding_dim = vision_tokenizer.dim_media
        self.num_tokens_per_vis = vision_tokenizer.num_tokens_per_media
        self.vision_encoder = vision_encoder
        self.vision_tokenizer = vision_tokenizer
        self.lang_model = lang_model
        if base_img_size is None:
            cfg = getattr(self.vision_encoder, "config", None)
            base_img_size = cfg.image_size if cfg is not None else self.vision_encoder.image_size[0]
        self.base_img_size = base_img_size
        self.pad_token_id = pad_token_id
        self.initial_tokenizer_len = initial_tokenizer_len
        std = getattr(self.lang_model.config, "initializer_range", 0.02)
        old_in = self.lang_model.get_input_embeddings()
        input_embeds = DecoupledEmbedding(max_original_id=initial_tokenizer_len - 1,
                                          num_additional_embeddings=len(self.special_tokens),
                                          _weight=old_in.weight, pad_token_id=self.pad_token_id)
        input_embeds.additional_embedding.to(old_in.weight.dtype).to(old_in.weight.device)
        input_embeds.additional_embedding.weight.data.normal_(mean=0.0, std=std)
        self.lang_model.set_input_embeddings(input_embeds)
        old_out = self.lang_model.get_output_embeddings()
        # NB: like the reference (src/vlm.py:88-93) this leaves DecoupledLinear's `bias=True` default in force even
        # when the LM head has no bias, so checkpoints carry lm_head.bias / lm_head.additional_fc.bias.
        out_embeds = DecoupledLinear(max_original_id=initial_tokenizer_len - 1,
                                     additional_out_features=len(self.special_tokens), _weight=old_out.weight,
                                     _bias=old_out.bias if hasattr(old_out, "bias") else None)
        out_embeds.to(old_out.weight.dtype).to(old_out.weight.device)
        out_embeds.additional_fc.to(old_out.weight.dtype).to(old_out.weight.device)
        out_embeds.additional_fc.weight.data.normal_(mean=0.0, std=std)
        self.lang_model.set_output_embeddings(out_embeds)
        self.vision_tokenizer._use_gradient_checkpointing = gradient_checkpointing

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

    # ---- bookkeeping the training scripts call --------------------------------------------------------------
    @property
    def num_trainable_params(self):
        return num_params(self, filter_to_trainable=True)

    def set_trainable(self):
        raise NotImplementedError

    def group_params_by_weight_decay(self):
        params_with_wd, params_without_wd = [], []
        for n, p in self.named_parameters():
            if p.requires_grad:
                (params_with_wd if self._should_apply_weight_decay(n) else params_without_wd).append(p)
        return params_with_wd, params_without_wd

    def _should_apply_weight_decay(self, parameter_name):
        raise NotImplementedError

    @property
    def special_tokens(self):
        assert "media_token" in self._special_tokens, \
            "VLMs need to request that the tokenizer add a media_token and call set_special_token_ids to set self.media_token_id"
        return self._special_tokens

    @property
    def special_token_ids(self):
        return [getattr(self, f"{att_name}_id") for att_name in self.special_tokens]

    def set_special_token_ids(self, string_to_ids):
        assert set(self.special_tokens.values()).issubset(set(string_to_ids.keys()))
        for att_name, token_str in self.special_tokens.items():
            token_id = string_to_ids[token_str]
            setattr(self, f"{att_name}_id", token_id)
            setattr(self.lang_model, f"{att_name}_id", token_id)

    def init_gradient_checkpointing(self):
        from torch.distributed.algorithms._checkpoint.checkpoint_wrapper import (
            checkpoint_wrapper, CheckpointWrapper, CheckpointImpl, apply_activation_checkpointing)
        from functools import partial
        wrapper = partial(checkpoint_wrapper, checkpoint_impl=CheckpointImpl.NO_REENTRANT)
        apply_activation_checkpointing(self, checkpoint_wrapper_fn=wrapper,
                                       check_fn=lambda m: getattr(m, "_use_gradient_checkpointing", False)
                                       and not isinstance(m, CheckpointWrapper))


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
                                    padding_side: str = "left", num_beams: int = 1):
        """src/vlm.py:445-603.  ``attention_mask`` in the returned dict is an ``ops.MaskTable``."""
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
                vision_tokens, self.media_token_id, self.pad_token_id, ASSISTANT_TOKEN_ID, padding_side, max_rects)
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

    def _post_forward_hook(self):
        pass

    def get_fsdp_lambda_fn(self):
        from torch.distributed.algorithms._checkpoint.checkpoint_wrapper import CheckpointWrapper
        decoder_block_class = getattr_recursive(self.lang_model, self.decoder_layers_attr_name)[0].__class__

        def lambda_fn(module: nn.Module):
            if getattr(module, "_use_gradient_checkpointing", False) and not isinstance(module, CheckpointWrapper):
                return False
            if module is self.vision_tokenizer:
                return True
            if isinstance(module, decoder_block_class):
                return True

        return lambda_fn

    def group_params_by_weight_decay(self):
        params_with_wd, params_without_wd = [], []
        for n, p in self.named_parameters():
            if p.requires_grad:
                (params_without_wd if "lang_model.model.embed_tokens" in n else params_with_wd).append(p)
        return params_with_wd, params_without_wd

    @property
    def num_params_per_module(self):
        return "\n".join([f"Vision encoder: {num_params(self.vision_encoder):,} parameters",
                          f"Vision tokenizer: {num_params(self.vision_tokenizer):,} parameters",
                          f"Language model: {num_params(self.lang_model):,} parameters"])

    @property
    def num_trainable_params_per_module(self):
        return "\n".join([
            f"Vision encoder: {num_params(self.vision_encoder, filter_to_trainable=True):,} trainable parameters",
            f"Vision tokenizer: {num_params(self.vision_tokenizer, filter_to_trainable=True):,} trainable parameters",
            f"Language model: {num_params(self.lang_model, filter_to_trainable=True):,} trainable parameters"])
