"""`create_model_and_transforms` (src/factory.py:21-159) on a tiny, locally saved SigLIP / Phi-3 / tokenizer triple: the same
call the reference's train.py:257-266 and eval.py:16-21 make, with `use_local_files=True`."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import load_golden
from golden import gen
from test_hub_twin import _tiny_checkpoints


def _tiny_tokenizer(path, n_words):
    """A WordLevel tokenizer of `n_words` entries saved in HF format (there is no network for the real Phi-3.5 tokenizer)."""
    from tokenizers import Tokenizer, models, pre_tokenizers
    from transformers import PreTrainedTokenizerFast
    vocab = {"<unk>": 0, "<s>": 1, "</s>": 2}
    for w in "what is in this image the a of and".split():
        vocab.setdefault(w, len(vocab))
    while len(vocab) < n_words:
        vocab[f"w{len(vocab)}"] = len(vocab)
    tk = Tokenizer(models.WordLevel(vocab, unk_token="<unk>"))
    tk.pre_tokenizer = pre_tokenizers.Whitespace()
    PreTrainedTokenizerFast(tokenizer_object=tk, unk_token="<unk>", bos_token="<s>", eos_token="</s>").save_pretrained(path)
    return path


def _create(tmp, device, dtype):
    from aki_amd import create_model_and_transforms
    T = gen.TINY
    _, _, pv, pl = _tiny_checkpoints(tmp)
    with open(os.path.join(pl, "generation_config.json"), "w") as f:      # what Phi-3.5-mini-instruct ships
        json.dump({"eos_token_id": [32007, 32001, 32000], "pad_token_id": 32000, "bos_token_id": 1}, f)
    ptok = _tiny_tokenizer(os.path.join(tmp, "tok"), T["vocab"] - 1)        # + <pad> = the tiny LM's vocabulary
    return create_model_and_transforms(clip_vision_encoder_path=pv, clip_vision_encoder_pretrained="google", lang_encoder_path=pl,
                                       tokenizer_path=ptok, use_local_files=True, verbose=False, device=device, dtype=dtype,
                                       n_px=T["image"], num_vision_tokens=T["num_vision_tokens"])


def test_create_model_and_transforms_builds_the_reference_module_tree(tmp_path):
    T = gen.TINY
    model, image_processor, tokenizer = _create(str(tmp_path), "cpu", torch.float32)
    # tokenizer: pad token added (the tiny tokenizer has none), then the two AKI tokens on top (src/factory.py:118-144)
    assert tokenizer.pad_token == "<pad>" and tokenizer.pad_token_id == T["vocab"] - 1
    assert len(tokenizer) == T["vocab"] + 2 and model.lang_model.config.vocab_size == len(tokenizer)
    assert model.media_token_id == tokenizer.convert_tokens_to_ids("<image>") == T["vocab"]
    assert model.end_of_trunk_token_id == tokenizer.convert_tokens_to_ids("<|endofchunk|>") == T["vocab"] + 1
    assert model.lang_model.media_token_id == model.media_token_id and model.pad_token_id == tokenizer.pad_token_id
    assert model.initial_tokenizer_len == T["vocab"] and model.decoder_layers_attr_name == "model.layers"
    # the module tree carries the reference's own state-dict keys and shapes (recorded from the imported reference)
    shapes = {k: tuple(s) for k, s in json.loads(str(load_golden("tiny_e2e.npz")["shapes"]))}
    assert {k: tuple(v.shape) for k, v in model.state_dict().items()} == shapes
    # trainable = everything but the vision tower (src/aki.py:52-57)
    assert not any(p.requires_grad for p in model.vision_encoder.parameters())
    assert all(p.requires_grad for p in model.vision_tokenizer.parameters()) and model.num_trainable_params > 0
    units = [m for m in model.modules() if model.get_fsdp_lambda_fn()(m)]
    assert len(units) == T["lm_layers"] + 1 and units[0] is model.vision_tokenizer
    decayed, plain = model.group_params_by_weight_decay()
    assert {id(p) for p in plain} == {id(p) for n, p in model.named_parameters() if "embed_tokens" in n and p.requires_grad}
    # generate() stops where HF's generate would: generation_config.json of the checkpoint
    assert model.default_eos_token_ids() == [32007, 32001, 32000]
    # image transform: resize to n_px, [0,1] -> [-1,1] (src/factory.py:79-84)
    from PIL import Image
    img = Image.fromarray((np.arange(40 * 30 * 3) % 255).astype(np.uint8).reshape(40, 30, 3))
    px = image_processor(img)
    assert px.shape == (3, T["image"], T["image"]) and px.dtype == torch.float32 and -1.0 <= float(px.min()) and float(px.max()) <= 1.0
    with pytest.raises(NotImplementedError):
        from aki_amd import create_model_and_transforms
        create_model_and_transforms("x", "openai", "y", "z")


@pytest.mark.gpu
def test_factory_model_runs_the_reference_call(tmp_path):
    """The object the factory returns answers the reference's training call `model(vision_x, lang_x, attention_mask, labels)[0]`
    (train/losses.py:110-115) on the GPU, and agrees with the hub twin built from the same checkpoints."""
    from aki_amd.modeling_aki import AKI as HubAKI
    T = gen.TINY
    model, image_processor, tokenizer = _create(str(tmp_path), "cuda", torch.float32)
    text = "<image> what is in this image"
    tokenizer.padding_side = "right"
    enc = tokenizer([text, text + " the a"], return_tensors="pt", padding=True)
    lang_x, am = enc["input_ids"].cuda(), enc["attention_mask"].cuda()
    assert int((lang_x == model.media_token_id).sum()) == 2
    vx = torch.randn(2, 1, 1, 3, T["image"], T["image"], generator=torch.Generator().manual_seed(3)).cuda()
    labels = lang_x.clone()
    labels[labels == tokenizer.pad_token_id] = -100
    labels[labels == model.media_token_id] = -100
    with torch.no_grad():
        out = model(vx, lang_x, attention_mask=am, labels=labels)
    L = lang_x.shape[1] - 1 + T["num_vision_tokens"]
    assert out.logits.shape == (2, L, len(tokenizer)) and bool(torch.isfinite(out.logits).all())
    assert out[0] is out.loss and float(out.loss) > 0
    twin = HubAKI(os.path.join(str(tmp_path), "vis"), os.path.join(str(tmp_path), "lm"), pad_token_id=tokenizer.pad_token_id,
                  initial_tokenizer_len=T["vocab"], tokenizer=tokenizer, num_vision_tokens=T["num_vision_tokens"]).cuda()
    twin.load_state_dict(model.state_dict())
    with torch.no_grad():
        out2 = twin(vx, lang_x, attention_mask=am, labels=labels)
    assert torch.equal(out.logits, out2.logits) and torch.equal(out.loss, out2.loss)
