"""The HF-Hub twin of the model class (src/modeling_aki.py; `AKI.from_pretrained(path, tokenizer=...)` in local_demo.py:30):
construction from checkpoint paths, the PyTorchModelHubMixin round trip and - on the GPU - the natively loaded Phi-3 and
SigLIP modules against the installed transformers modules they were loaded from."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import load_golden
from golden import gen


def _tiny_checkpoints(tmp):
    from transformers import Phi3Config, Phi3ForCausalLM, SiglipConfig, SiglipModel, SiglipTextConfig, SiglipVisionConfig
    T = gen.TINY
    torch.manual_seed(0)
    lm_cfg = Phi3Config(vocab_size=T["vocab"], hidden_size=T["lm_hidden"], intermediate_size=T["lm_inter"],
                        num_hidden_layers=T["lm_layers"], num_attention_heads=T["lm_heads"], num_key_value_heads=T["lm_heads"],
                        max_position_embeddings=4096, original_max_position_embeddings=4096, pad_token_id=T["pad_token_id"],
                        attn_implementation="eager")
    lm = Phi3ForCausalLM(lm_cfg)
    lm.save_pretrained(os.path.join(tmp, "lm"))
    sc = SiglipConfig(text_config=SiglipTextConfig(hidden_size=32, intermediate_size=64, num_hidden_layers=1, num_attention_heads=2,
                                                   vocab_size=100, bos_token_id=1, eos_token_id=2).to_dict(),
                      vision_config=SiglipVisionConfig(hidden_size=T["vis_hidden"], intermediate_size=T["vis_inter"],
                                                       num_hidden_layers=T["vis_layers"], num_attention_heads=T["vis_heads"],
                                                       image_size=T["image"], patch_size=T["patch"], attn_implementation="eager").to_dict())
    vis = SiglipModel(sc)
    vis.save_pretrained(os.path.join(tmp, "vis"))
    return lm, vis, os.path.join(tmp, "vis"), os.path.join(tmp, "lm")


def test_hub_twin_constructs_from_paths_and_round_trips(tmp_path):
    from aki_amd.modeling_aki import AKI
    T = gen.TINY
    _, _, pv, pl = _tiny_checkpoints(str(tmp_path))
    m = AKI(pv, pl, pad_token_id=T["pad_token_id"], initial_tokenizer_len=T["vocab"], num_vision_tokens=T["num_vision_tokens"])
    # the reference's own key set and shapes (recorded from the imported reference by make_golden.py)
    shapes = {k: tuple(s) for k, s in json.loads(str(load_golden("tiny_e2e.npz")["shapes"]))}
    assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == shapes
    m.set_special_token_ids({"<image>": T["media_token_id"], "<|endofchunk|>": T["eoc_token_id"]})
    m.save_pretrained(str(tmp_path / "aki"))
    assert {"config.json", "model.safetensors"} <= set(os.listdir(tmp_path / "aki"))
    m2 = AKI.from_pretrained(str(tmp_path / "aki"))
    for (ka, a), (kb, b) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert ka == kb and torch.equal(a, b)
    assert hasattr(m2, "generate") and m2.num_tokens_per_vis == T["num_vision_tokens"]


@pytest.mark.gpu
def test_natively_loaded_modules_match_transformers(tmp_path):
    """Phi-3 stack and SigLIP tower loaded from HF checkpoints: exact-f32 HIP forward == the transformers modules' own eager
    forward on the CPU (plain causal attention for the LM; the third-party code the reference delegates to)."""
    from aki_amd.modeling_aki import native_language_model, native_vision_tower
    hf_lm, hf_vis, pv, pl = _tiny_checkpoints(str(tmp_path))
    T = gen.TINY
    ids = torch.randint(3, 31000, (2, 37), generator=torch.Generator().manual_seed(1))
    with torch.no_grad():
        want = hf_lm.eval()(input_ids=ids).logits
        lm = native_language_model(pl).to("cuda").eval()
        got = lm(input_ids=ids.cuda()).logits.cpu()
        assert (got - want).abs().max().item() <= 1e-5 * max(1.0, want.abs().max().item())
        px = torch.randn(2, 3, T["image"], T["image"], generator=torch.Generator().manual_seed(2))
        wantv = hf_vis.eval().vision_model(pixel_values=px).last_hidden_state
        vt = native_vision_tower(pv).to("cuda").eval()
        gotv = vt(px.cuda()).last_hidden_state.cpu()
        assert (gotv - wantv).abs().max().item() <= 1e-5 * max(1.0, wantv.abs().max().item())
