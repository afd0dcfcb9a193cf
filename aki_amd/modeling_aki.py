"""HF-Hub twin of the AKI model class: host-side mirror of src/modeling_aki.py (the class `local_demo.py:30` and
`eval_cv_bench/eval.py:16-21` obtain through ``AKI.from_pretrained(path, tokenizer=...)``).

Same constructor signature as the reference (paths of the vision tower and the language model instead of module objects)
and the same ``PyTorchModelHubMixin`` behaviour: ``from_pretrained`` builds the module tree from the stored constructor
arguments and loads ``model.safetensors`` into it - the state-dict keys of this class are the reference's keys, so a
checkpoint written by the reference loads unchanged.  The modules themselves are the MI355X-native ones: the HF modules
the two paths resolve to are only read for their config and weights.

Reference: /root/reference/codes/open_flamingo/src/modeling_aki.py  (__init__ :12-75, forward :83-151, generate :153-226)
"""
from __future__ import annotations

from typing import Optional

import torch
from huggingface_hub import PyTorchModelHubMixin

from .aki import AKI as _TrainTimeAKI
from .helpers import PerceiverResampler
from .phi3 import Phi3ForCausalLM
from .siglip import SiglipVisionTransformer
from .vlm import VLMWithLanguageStream


def native_vision_tower(vision_encoder_path: str, **hf_kwargs) -> SiglipVisionTransformer:
    """``AutoModel.from_pretrained(path).vision_model`` (src/modeling_aki.py:37-38) -> native SigLIP tower with its weights."""
    from transformers import AutoModel
    hf_vis = AutoModel.from_pretrained(vision_encoder_path, **hf_kwargs).vision_model
    vt = SiglipVisionTransformer(hf_vis.config)
    missing = vt.load_state_dict(hf_vis.state_dict(), strict=False)      # the pooling head is not used by AKI
    assert not missing.missing_keys, missing.missing_keys
    return vt


def native_language_model(lang_model_path: str, **hf_kwargs) -> Phi3ForCausalLM:
    """``AutoModelForCausalLM.from_pretrained(path, trust_remote_code=True)`` (src/modeling_aki.py:42-46) -> native Phi-3."""
    from transformers import AutoModelForCausalLM
    hf_lm = AutoModelForCausalLM.from_pretrained(lang_model_path, trust_remote_code=True, **hf_kwargs)
    lm = Phi3ForCausalLM(hf_lm.config)
    lm.load_state_dict(hf_lm.state_dict(), strict=True)
    lm.generation_config = getattr(hf_lm, "generation_config", None)
    return lm


class AKI(VLMWithLanguageStream, PyTorchModelHubMixin):
    def __init__(self, vision_encoder_path: str, lang_model_path: str, pad_token_id: int, initial_tokenizer_len: Optional[int] = None,
                 tokenizer=None, decoder_layers_attr_name: str = None, gradient_checkpointing: bool = False,
                 base_img_size: Optional[int] = None, num_vision_tokens: int = 144):
        vision_encoder = native_vision_tower(vision_encoder_path)
        vis_feature_dim = vision_encoder.config.hidden_size
        lang_model = native_language_model(lang_model_path, local_files_only=False)
        self._special_tokens = {"media_token": "<image>", "end_of_trunk_token": "<|endofchunk|>"}
        lang_embedding_dim = lang_model.get_input_embeddings().weight.shape[1]
        if decoder_layers_attr_name is None:
            decoder_layers_attr_name = "model.layers"
        super().__init__(
            vision_encoder=vision_encoder,
            vision_tokenizer=PerceiverResampler(dim=vis_feature_dim, dim_inner=lang_embedding_dim, num_latents=num_vision_tokens),
            lang_model=lang_model, initial_tokenizer_len=initial_tokenizer_len, gradient_checkpointing=gradient_checkpointing,
            base_img_size=base_img_size, decoder_layers_attr_name=decoder_layers_attr_name, pad_token_id=pad_token_id)
        if tokenizer is not None:
            self.lang_model.config.vocab_size = len(tokenizer)
            self.set_special_token_ids({v: tokenizer.convert_tokens_to_ids(v) for v in self.special_tokens.values()})

    set_trainable = _TrainTimeAKI.set_trainable
    default_eos_token_ids = _TrainTimeAKI.default_eos_token_ids
    _should_apply_weight_decay = _TrainTimeAKI._should_apply_weight_decay
    forward = _TrainTimeAKI.forward          # src/modeling_aki.py:83-151 is identical to src/aki.py:65-134
    generate = _TrainTimeAKI.generate        # MMA prefill + HIP decode steps (aki_amd/aki.py)
    _beam_search = _TrainTimeAKI._beam_search
