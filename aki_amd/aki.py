"""AKI model class: host-side mirror of src/aki.py (train-time AKI) with the same constructor and
forward signature; see also src/modeling_aki.py:83-151 (HF-Hub twin, same forward).

Reference: /root/reference/codes/open_flamingo/src/aki.py  (__init__ :10-50, set_trainable :52-57, forward :65-134)
"""
from __future__ import annotations

from typing import List, Optional, Tuple, Union

import torch
from torch import nn

from .helpers import PerceiverResampler
from .vlm import VLMWithLanguageStream


class AKI(VLMWithLanguageStream):
    def __init__(self, vision_encoder: nn.Module, lang_model: nn.Module, vis_feature_dim: int, initial_tokenizer_len: int,
                 pad_token_id: int, decoder_layers_attr_name: str = None, gradient_checkpointing: bool = False,
                 base_img_size: Optional[int] = None, num_vision_tokens: int = 144):
        self._special_tokens = {"media_token": "<image>", "end_of_trunk_token": "<|endofchunk|>"}
        lang_embedding_dim = lang_model.get_input_embeddings().weight.shape[1]
        super().__init__(
            vision_encoder=vision_encoder,
            vision_tokenizer=PerceiverResampler(dim=vis_feature_dim, dim_inner=lang_embedding_dim, num_latents=num_vision_tokens),
            lang_model=lang_model, initial_tokenizer_len=initial_tokenizer_len, gradient_checkpointing=gradient_checkpointing,
            base_img_size=base_img_size, decoder_layers_attr_name=decoder_layers_attr_name, pad_token_id=pad_token_id)

    def set_trainable(self):
        """Unfreeze everything except the vision_encoder (src/aki.py:52-57)."""
        self.requires_grad_(True)
        self.vision_encoder.requires_grad_(False)

    def _should_apply_weight_decay(self, parameter_name):
        return True

    def forward(self, vision_x: Optional[torch.Tensor], lang_x: torch.Tensor, attention_mask: Optional[torch.Tensor] = None,
                labels: Optional[torch.Tensor] = None, image_size: Optional[Tuple] = None,
                past_key_values: Optional[List[Union[torch.Tensor, Tuple[torch.Tensor]]]] = None,
                past_media_locations: Optional[torch.Tensor] = None, past_vision_tokens: Optional[torch.Tensor] = None,
                use_cache: Optional[bool] = False, **kwargs):
        """vision_x (B, T_img, F=1, C, H, W); lang_x (B, T_txt) with <image> placeholders -> CausalLMOutputWithPast
        whose logits cover the EXPANDED stream length L (src/aki.py:65-134)."""
        assert not (past_vision_tokens is None) ^ (past_media_locations is None), \
            "past_vision_tokens and past_media_locations must both be None or both be not None"
        if vision_x is not None:
            vision_tokens = self.vision_tokenizer(self._encode_vision_x(vision_x=vision_x))
        else:
            vision_tokens = None
        new_inputs = self._prepare_inputs_for_forward(
            vision_tokens=vision_tokens, lang_x=lang_x, attention_mask=attention_mask, vision_attention_mask=None,
            labels=labels, past_key_values=past_key_values, past_media_locations=past_media_locations,
            padding_side="right", past_vision_tokens=past_vision_tokens)
        output = self.lang_model(**new_inputs, use_cache=use_cache, past_key_values=past_key_values, **kwargs)
        self._post_forward_hook()
        return output

    def generate(self, vision_x, lang_x, image_size=None, attention_mask=None, past_key_values=None,
                 past_media_locations=None, past_vision_tokens=None, **kwargs):
        raise NotImplementedError("generate() needs the KV-cache decode path: SURVEY 8(f) item 1, next after the forward pass")
