"""create_model_and_transforms: host-side mirror of src/factory.py:21-159 for the AKI configuration
(SigLIP vision tower + Phi-3(.5)-mini), plus ``build_aki`` for synthetic / random-init models.

The reference loads HF checkpoints with AutoModel / AutoModelForCausalLM and wraps them.  Here the HF
objects are only used as *containers of weights and configs*: their state dicts are loaded into the
MI355X-native module tree (same parameter names), which executes on the HIP kernels.
"""
from __future__ import annotations

from typing import Optional

import torch

from .aki import AKI
from .phi3 import Phi3ForCausalLM, make_phi3_config
from .siglip import SiglipVisionTransformer, make_siglip_config

__KNOWN_DECODER_LAYERS_ATTR_NAMES = {"phi": "model.layers", "llama": "model.layers", "mistral": "model.layers"}


def _infer_decoder_layers_attr_name(model):
    for k, v in __KNOWN_DECODER_LAYERS_ATTR_NAMES.items():
        if k.lower() in model.__class__.__name__.lower():
            return v
    raise ValueError("We require the attribute name for the nn.ModuleList in the decoder storing the transformer block "
                     "layers. Please supply this string manually.")


def build_aki(lm_config=None, vis_config=None, initial_tokenizer_len: int = 32011, pad_token_id: int = 32000,
              num_vision_tokens: int = 144, dtype=torch.bfloat16, device="cuda", init_std: Optional[float] = None,
              seed: int = 0, media_token_id: Optional[int] = None, gradient_checkpointing: bool = False):
    """Random-init AKI of a given shape (defaults = AKI-4B: Phi-3.5-mini + SigLIP-so400m/14-384 + 144 latents).
    Weights ~ N(0, initializer_range) with unit norm gains, generated directly on the device."""
    lm_config = lm_config or make_phi3_config()
    vis_config = vis_config or make_siglip_config()
    with torch.device("meta"):
        lm = Phi3ForCausalLM(lm_config)
        vt = SiglipVisionTransformer(vis_config)
    lm = lm.to_empty(device=device).to(dtype)
    vt = vt.to_empty(device=device).to(dtype)
    model = AKI(vision_encoder=vt, lang_model=lm, vis_feature_dim=vis_config.hidden_size,
                initial_tokenizer_len=initial_tokenizer_len, pad_token_id=pad_token_id,
                decoder_layers_attr_name="model.layers", num_vision_tokens=num_vision_tokens,
                gradient_checkpointing=gradient_checkpointing)
    model = model.to(device=device, dtype=dtype)
    std = init_std if init_std is not None else getattr(lm_config, "initializer_range", 0.02)
    g = torch.Generator(device=device).manual_seed(seed)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if p.dim() == 1 and n.endswith("weight"):
                p.fill_(1.0)
            elif n.endswith("bias"):
                p.zero_()
            elif n.endswith("latents"):
                p.copy_(torch.randn(p.shape, generator=g, device=device, dtype=torch.float32))
            else:
                p.copy_((torch.randn(p.shape, generator=g, device=device, dtype=torch.float32) * std))
    model.lang_model.config.vocab_size = initial_tokenizer_len + len(model.special_tokens)   # src/factory.py:144
    mid = media_token_id if media_token_id is not None else initial_tokenizer_len
    model.set_special_token_ids({"<image>": mid, "<|endofchunk|>": mid + 1})
    model.set_trainable()
    return model


def _siglip_image_processor(n_px: int):
    """src/factory.py:79-84 without torchvision: bicubic resize to n_px, to tensor, normalise with mean=std=0.5."""
    def proc(img):
        import numpy as np
        from PIL import Image
        img = img.convert("RGB").resize((n_px, n_px), resample=Image.BICUBIC)
        x = torch.from_numpy(np.asarray(img, dtype=np.float32) / 255.0).permute(2, 0, 1)
        return (x - 0.5) / 0.5
    return proc


def create_model_and_transforms(clip_vision_encoder_path: str, clip_vision_encoder_pretrained: str, lang_encoder_path: str,
                                tokenizer_path: str, use_local_files: bool = False, decoder_layers_attr_name: str = None,
                                cache_dir: Optional[str] = None, gradient_checkpointing: bool = False, verbose: bool = True,
                                dtype=torch.bfloat16, device="cuda", **model_kwargs):
    """Same signature and return triple as the reference.  Only the AKI configuration
    (``clip_vision_encoder_pretrained == 'google'`` SigLIP + a Phi-3 family LM) is on the MI355X path."""
    from transformers import AutoConfig, AutoModel, AutoModelForCausalLM, AutoTokenizer
    if clip_vision_encoder_pretrained != "google":
        raise NotImplementedError("the MI355X path implements AKI's SigLIP tower (clip_vision_encoder_pretrained='google'); "
                                  "open_clip / OpenAI CLIP towers are outside the hot path (SURVEY 2, row 6)")
    n_px = model_kwargs.pop("n_px", 384)
    hf_vis = AutoModel.from_pretrained(clip_vision_encoder_path, local_files_only=use_local_files, cache_dir=cache_dir).vision_model
    vis_cfg = hf_vis.config
    vt = SiglipVisionTransformer(vis_cfg)
    missing = vt.load_state_dict(hf_vis.state_dict(), strict=False)      # the pooling head is not used by AKI
    assert not missing.missing_keys, missing.missing_keys
    image_processor = _siglip_image_processor(n_px)
    text_tokenizer = AutoTokenizer.from_pretrained(tokenizer_path, local_files_only=use_local_files, trust_remote_code=True,
                                                   cache_dir=cache_dir, use_fast=False)
    text_tokenizer.add_bos_token = True
    text_tokenizer.add_eos_token = True
    hf_lm = AutoModelForCausalLM.from_pretrained(lang_encoder_path, local_files_only=use_local_files, trust_remote_code=True,
                                                 cache_dir=cache_dir)
    lm = Phi3ForCausalLM(hf_lm.config)
    lm.load_state_dict(hf_lm.state_dict(), strict=True)
    lm.generation_config = getattr(hf_lm, "generation_config", None)
    del hf_lm, hf_vis
    if text_tokenizer.pad_token is None or text_tokenizer.pad_token == text_tokenizer.eos_token:
        text_tokenizer.add_special_tokens({"pad_token": "<pad>"})
    if decoder_layers_attr_name is None:
        decoder_layers_attr_name = _infer_decoder_layers_attr_name(lm)
    model = AKI(vision_encoder=vt.to(device=device, dtype=dtype), lang_model=lm.to(device=device, dtype=dtype),
                vis_feature_dim=vis_cfg.hidden_size, initial_tokenizer_len=len(text_tokenizer),
                gradient_checkpointing=gradient_checkpointing, decoder_layers_attr_name=decoder_layers_attr_name,
                pad_token_id=text_tokenizer.pad_token_id, **model_kwargs).to(device=device, dtype=dtype)
    text_tokenizer.add_special_tokens({"additional_special_tokens": list(model.special_tokens.values())})
    model.lang_model.config.vocab_size = len(text_tokenizer)
    model.set_special_token_ids({v: text_tokenizer.convert_tokens_to_ids(v) for v in model.special_tokens.values()})
    model.set_trainable()
    if verbose:
        print(f"==========Model initialized with {model.num_trainable_params:,} trainable parameters")
        print(f"==========Trainable Parameters\n{model.num_trainable_params_per_module}")
        print(f"==========Total Parameters\n{model.num_params_per_module}\n==========")
    return model, image_processor, text_tokenizer
