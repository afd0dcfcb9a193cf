"""End-to-end GPU parity: aki_amd.AKI (HIP kernels behind the reference's model API) against golden outputs of the
reference's own AKI.forward on the same weights and inputs (tests/golden/make_golden.py)."""
import json

import numpy as np
import pytest
import torch

from conftest import load_golden
from golden import gen
import aki_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


def build_tiny(dtype):
    from aki_amd.factory import build_aki
    from aki_amd.phi3 import make_phi3_config
    from aki_amd.siglip import make_siglip_config
    T = gen.TINY
    m = build_aki(make_phi3_config(vocab_size=T["vocab"], hidden_size=T["lm_hidden"], intermediate_size=T["lm_inter"],
                                   num_hidden_layers=T["lm_layers"], num_attention_heads=T["lm_heads"],
                                   num_key_value_heads=T["lm_heads"], pad_token_id=T["pad_token_id"]),
                  make_siglip_config(hidden_size=T["vis_hidden"], intermediate_size=T["vis_inter"],
                                     num_hidden_layers=T["vis_layers"], num_attention_heads=T["vis_heads"],
                                     image_size=T["image"], patch_size=T["patch"]),
                  initial_tokenizer_len=T["vocab"], pad_token_id=T["pad_token_id"], num_vision_tokens=T["num_vision_tokens"],
                  dtype=dtype, device=DEV)
    g = load_golden("tiny_e2e.npz")
    shapes = [(k, tuple(s)) for k, s in json.loads(str(g["shapes"]))]
    # the native module tree must expose exactly the reference's state-dict keys and shapes
    mine = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    assert mine == dict(shapes)
    p = gen.fill_params(shapes, 11)
    m.load_state_dict({k: torch.from_numpy(v).to(dtype) for k, v in p.items()}, strict=True)
    m.eval()
    return m, g


def batch(g, dtype):
    rng = gen.rng_for("tiny_batch")
    B = g["lang_x"].shape[0]
    vx = rng.standard_normal((B, 1, 1, 3, gen.TINY["image"], gen.TINY["image"]), dtype=np.float32)
    return (torch.from_numpy(vx).to(DEV).to(dtype), torch.from_numpy(g["lang_x"]).to(DEV),
            torch.from_numpy(g["attention_mask"]).to(DEV), torch.from_numpy(g["labels"]).to(DEV))


def test_tiny_aki_forward_fp32_vs_reference():
    m, g = build_tiny(torch.float32)
    vx, lx, am, lab = batch(g, torch.float32)
    with torch.no_grad():
        feats = m._encode_vision_x(vx)
        vtok = m.vision_tokenizer(feats)
        prep = m._prepare_inputs_for_forward(vision_tokens=vtok, lang_x=lx, attention_mask=am, labels=lab, padding_side="right")
        out = m(vx, lx, attention_mask=am, labels=lab)
    for got_, want_, what_ in ((feats, g["vision_feats"], "SigLIP features"), (vtok, g["vision_tokens"], "vision tokens"),
                               (prep["inputs_embeds"], g["inputs_embeds"], "spliced inputs_embeds")):
        e_ = np.abs(got_.cpu().numpy() - want_)
        assert e_.max() <= 1e-5 * max(1.0, np.abs(want_).max()), f"fp32 {what_}: max err {e_.max():.3g} (max |ref| {np.abs(want_).max():.3g})"
    assert np.array_equal(prep["labels"].cpu().numpy(), g["new_labels"])
    from aki_amd import ops
    dense = ops.mask_dense(prep["attention_mask"], lx.shape[0]).cpu().numpy()
    assert np.array_equal(dense, gen.unpack_mask_bits(g["mask_bits"], tuple(g["mask_shape"])))
    logits = out.logits[:, :, torch.from_numpy(g["logit_cols"]).to(DEV)].cpu().numpy()
    assert out.logits.shape[-1] == gen.TINY["vocab"] + 2
    err = np.abs(logits - g["logits"])
    from conftest import record_parity
    record_parity("tiny AKI end to end, fp32 logits vs the reference's own run", torch.float32, err.max(), err.mean(), np.abs(g["logits"]).max(),
                  "1e-5*max(1,max|ref|)")
    assert err.max() <= 1e-5 * max(1.0, np.abs(g["logits"]).max()), f"fp32 logits: max err {err.max():.3g}"
    assert abs(float(out.loss) - float(g["loss"])) < 1e-4
    assert abs(float(out[0]) - float(g["loss"])) < 1e-4            # train/losses.py:110-115 uses model(...)[0]


def test_tiny_aki_forward_bf16_vs_reference():
    m, g = build_tiny(torch.bfloat16)
    vx, lx, am, lab = batch(g, torch.bfloat16)
    with torch.no_grad():
        out = m(vx, lx, attention_mask=am, labels=lab)
    logits = out.logits[:, :, torch.from_numpy(g["logit_cols"]).to(DEV)].float().cpu().numpy()
    assert np.isfinite(logits).all()
    e_hip = np.abs(logits - g["logits"])
    e_ref = np.abs(g["logits16"] - g["logits"])          # the reference's own bf16 eager path vs its fp32 path
    # non-pad rows only matter downstream, but every row is checked: pad rows reproduce the uniform-softmax convention
    assert e_hip.mean() <= 1.5 * e_ref.mean() + 1e-3, f"bf16 mean |err| {e_hip.mean():.4g} vs reference bf16 eager {e_ref.mean():.4g}"
    assert e_hip.max() <= 2.0 * e_ref.max() + 1e-2, f"bf16 max |err| {e_hip.max():.4g} vs reference bf16 eager {e_ref.max():.4g}"
    assert abs(float(out.loss) - float(g["loss"])) < 2e-2


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_lang_model_accepts_the_reference_dense_mask_bit_exact(dtype):
    """The reference hands `attention_mask (B,1,L,L) int64 0/1` to lang_model (src/vlm.py:589-603 -> src/aki.py:125-130).
    Passing that dense tensor - here the reference's OWN mask from the golden file - must give exactly the logits of the
    MaskTable path: it is converted on the device (aki_mma_mask_to_table), verified, and runs the same kernels."""
    from aki_amd import ops
    m, g = build_tiny(dtype)
    vx, lx, am, lab = batch(g, dtype)
    B = lx.shape[0]
    with torch.no_grad():
        prep = m._prepare_inputs_for_forward(vision_tokens=m.vision_tokenizer(m._encode_vision_x(vx)), lang_x=lx,
                                             attention_mask=am, labels=lab, padding_side="right")
        dense = torch.from_numpy(gen.unpack_mask_bits(g["mask_bits"], tuple(g["mask_shape"])).astype(np.int64)).to(DEV)
        assert dense.shape == (B, 1, prep["inputs_embeds"].shape[1], prep["inputs_embeds"].shape[1])
        via_table = m.lang_model(inputs_embeds=prep["inputs_embeds"], attention_mask=prep["attention_mask"], labels=prep["labels"])
        via_dense = m.lang_model(inputs_embeds=prep["inputs_embeds"], attention_mask=dense, labels=prep["labels"])
    assert torch.equal(via_table.logits, via_dense.logits)
    assert torch.equal(via_table.loss, via_dense.loss)
    got = ops.mask_to_table(dense)
    want = prep["attention_mask"]
    assert torch.equal(got.col_valid_bits, want.col_valid_bits) and torch.equal(got.seq_lens, want.seq_lens)
    # KV-cache prefill through the dense mask, too
    with torch.no_grad():
        c1 = m.lang_model(inputs_embeds=prep["inputs_embeds"], attention_mask=dense, use_cache=True).past_key_values
        c2 = m.lang_model(inputs_embeds=prep["inputs_embeds"], attention_mask=want, use_cache=True).past_key_values
    assert torch.equal(c1.cache_len, c2.cache_len) and torch.equal(c1.k[1][:, :, :4], c2.k[1][:, :, :4])
    # what the family does not contain is refused, never approximated
    bad = dense.clone()
    bad[0, 0, 5, 2] = 0
    with torch.no_grad(), pytest.raises(ops.AkiError):
        m.lang_model(inputs_embeds=prep["inputs_embeds"], attention_mask=bad)
    with torch.no_grad(), pytest.raises(ValueError):
        m.lang_model(inputs_embeds=prep["inputs_embeds"], attention_mask=(1.0 - dense.float()) * -1e9)


def test_reference_api_behaviour():
    m, g = build_tiny(torch.float32)
    vx, lx, am, lab = batch(g, torch.float32)
    # multi-image prompts raise like the reference unless the build-defined extension is switched on
    lx2 = lx.clone()
    lx2[0, 2] = gen.TINY["media_token_id"]
    vx2 = torch.cat([vx, vx], dim=1)
    with torch.no_grad(), pytest.raises(RuntimeError):
        m(vx2, lx2, attention_mask=am)
    m.allow_multi_image = True
    with torch.no_grad():
        out = m(vx2, lx2, attention_mask=am)
    assert out.logits.shape[1] == lx.shape[1] + 2 * (gen.TINY["num_vision_tokens"] - 1)
    # dense-mask compatibility entry point is bit-exact with the reference's
    gm = load_golden("mask_cases.npz")
    am0, s, t, e = gen.mask_cases()[0]
    dm = m._make_modality_mutual_mask(torch.from_numpy(am0).to(DEV), s, t, e, torch.Size([len(am0)]), torch.int64, DEV)
    assert dm.dtype == torch.int64 and np.array_equal(dm.cpu().numpy(), gen.unpack_mask_bits(gm["bits_0"], (1, len(am0), len(am0))))
    # bookkeeping used by train.py
    assert m.num_trainable_params > 0 and not any(p.requires_grad for p in m.vision_encoder.parameters())
    wd, nwd = m.group_params_by_weight_decay()
    assert len(nwd) >= 1 and len(wd) > len(nwd)
    with pytest.raises(NotImplementedError):
        m.generate(vx, lx, do_sample=True, num_beams=2)          # beam-sample decoding is not implemented


def test_cpu_tensors_fail_loudly():
    from aki_amd import AkiError
    m, g = build_tiny(torch.float32)
    vx, lx, am, lab = batch(g, torch.float32)
    with pytest.raises((AkiError, RuntimeError)):
        m.cpu()(vx.cpu(), lx.cpu(), attention_mask=am.cpu())


def test_config1_full_width_fp32_vs_oracle():
    """BASELINE configs[0] - one image + 64-token prompt, batch 1, fp32 (L = 207) - at the FULL WIDTH of AKI-4B (d 3072,
    32 heads x 96, FFN 8192, vocab 32011+2, SigLIP 1152 / 16 heads / MLP 4304 at 384 px = 729 patches, Perceiver 6 layers
    with 144 latents) and reduced depth (2 decoder + 2 SigLIP layers, so the fp32 weights fit the test budget): the exact-f32
    HIP path against the torch oracle (oracle/aki_torch.py, pinned to the reference) on the same weights."""
    import aki_torch as OT
    from aki_amd.factory import build_aki
    from aki_amd.phi3 import make_phi3_config
    from aki_amd.siglip import make_siglip_config
    m = build_aki(make_phi3_config(num_hidden_layers=2), make_siglip_config(num_hidden_layers=2), dtype=torch.float32, device=DEV, seed=3)
    m.eval()
    g = torch.Generator().manual_seed(5)
    N_TXT = 64
    ids = torch.randint(3, 31999, (1, N_TXT), generator=g)
    ids[0, 0], ids[0, 6], ids[0, N_TXT - 17], ids[0, -1] = 1, m.media_token_id, 32001, 2
    am = torch.ones_like(ids)
    labels = ids.clone()
    labels[labels == m.media_token_id] = -100
    vx = torch.rand(1, 1, 1, 3, 384, 384, generator=g) * 2 - 1
    with torch.no_grad():
        out = m(vx.to(DEV), ids.to(DEV), attention_mask=am.to(DEV), labels=labels.to(DEV))
    assert out.logits.shape == (1, N_TXT - 1 + 144, 32011 + 2)
    p = {k: v.detach().float().cpu() for k, v in m.state_dict().items()}
    cfg = dict(vis_layers=2, vis_heads=16, lm_layers=2, lm_heads=32, max_original_id=32010, media_token_id=m.media_token_id,
               pad_token_id=32000, num_vision_tokens=144)
    torch.set_num_threads(min(16, torch.get_num_threads()))
    with torch.no_grad():
        ref = OT.aki_forward(p, cfg, vx, ids, am, labels)
    cols = torch.cat([torch.arange(0, 32013, 499), torch.tensor([1, 2, 32000, 32001, 32011, 32012])])
    got, want = out.logits[0][:, cols.to(DEV)].cpu(), ref["logits"][0][:, cols]
    err = (got - want).abs().max().item()
    from conftest import record_parity
    record_parity("AKI-4B width, 2+2 layers, fp32 logits vs the torch oracle", torch.float32, err, (got - want).abs().mean().item(),
                  want.abs().max().item(), "1e-5*max(1,max|ref|)")
    assert err <= 1e-5 * max(1.0, want.abs().max().item()), f"fp32 logits at full width: max err {err:.3g} (max |ref| {want.abs().max().item():.3g})"
    assert abs(float(out.loss) - float(ref["loss"])) < 1e-4


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_full_width_decoder_vs_transformers_eager_under_441_mask(dtype):
    """The arithmetic the reference delegates to (transformers' Phi3 eager attention, HF:phi3/modeling_phi3.py:145-263) under
    the transformers==4.41.2 mask hand-off (dense 0/1 MMA mask -> 1 - mask -> finfo.min), at the FULL WIDTH of Phi-3.5-mini
    (d 3072, 32 heads x 96, FFN 8192, vocab 32064; 2 layers): HF runs on the CPU with the dense mask materialised from the
    MaskTable (bit-exact vs the reference, test_mask_dense_all_reference_cases), the HIP path consumes the table itself.
    fp32: logits within the stated 1e-5 of max|ref|; bf16: HIP error vs HF-fp32 no larger than 1.5x HF's own bf16-eager error."""
    from transformers import Phi3Config, Phi3ForCausalLM as HFPhi3
    from aki_amd import ops
    from aki_amd.phi3 import Phi3ForCausalLM
    torch.manual_seed(0)
    cfg = Phi3Config(vocab_size=32064, hidden_size=3072, intermediate_size=8192, num_hidden_layers=2, num_attention_heads=32,
                     num_key_value_heads=32, max_position_embeddings=4096, original_max_position_embeddings=4096,
                     pad_token_id=32000, attn_implementation="eager")
    hf = HFPhi3(cfg).eval()
    from conftest import randomize_norms_and_biases, record_parity
    assert randomize_norms_and_biases(hf, seed=21) == 5         # RMSNorm gains 1 + 0.2 N: the gain fold is not the identity
    B, L = 2, 300
    x = torch.randn(B, L, 3072, generator=torch.Generator().manual_seed(1)) * 0.5
    am = np.ones((B, L), dtype=bool)
    am[1, 260:] = False
    rects = [[(6, 150, 150, 283)], [(10, 154, 154, 240)]]
    table = ops.MaskTable.from_host(rects, am, [L, L], DEV)
    dense = ops.mask_dense(table, B).cpu()                                   # (B,1,L,L) int64 0/1, the reference's tensor
    inv = 1.0 - dense.float()
    add_mask = inv.masked_fill(inv.bool(), torch.finfo(torch.float32).min)  # transformers==4.41.2 _prepare_4d_causal_attention_mask
    pos = torch.arange(L)[None]
    valid = torch.from_numpy(am)[..., None]
    with torch.no_grad():
        want = hf(inputs_embeds=x, attention_mask=add_mask, position_ids=pos).logits
        lm = Phi3ForCausalLM(cfg)
        lm.load_state_dict(hf.state_dict(), strict=True)
        lm = lm.to(DEV).to(dtype).eval()
        got = lm(inputs_embeds=x.to(DEV).to(dtype), attention_mask=table).logits.float().cpu()
        if dtype == torch.float32:
            err = ((got - want).abs() * valid).max().item()
            record_parity("full-width decoder, fp32 logits vs transformers eager", torch.float32, err, ((got - want).abs() * valid).mean().item(),
                          want.abs().max().item(), "1e-5*max(1,max|ref|)")
            assert err <= 1e-5 * max(1.0, want.abs().max().item()), err
            # padded rows (all-zero mask rows): uniform softmax under finfo.min, reproduced by the kernel
            errp = ((got - want).abs() * (~valid)).max().item()
            assert errp <= 1e-3 * max(1.0, want.abs().max().item()), errp
        else:
            lm.model.fold_norms = False                      # every RMSNorm as its own kernel: recorded beside the folded path
            got_unf = lm(inputs_embeds=x.to(DEV).to(dtype), attention_mask=table).logits.float().cpu()
            assert not torch.equal(got, got_unf)
            hf16 = hf.to(torch.bfloat16)
            m16 = add_mask.to(torch.bfloat16).masked_fill(inv.bool(), torch.finfo(torch.bfloat16).min)
            ref16 = hf16(inputs_embeds=x.to(torch.bfloat16), attention_mask=m16, position_ids=pos).logits.float()
            e_hip = ((got - want).abs() * valid).mean().item()
            e_unf = ((got_unf - want).abs() * valid).mean().item()
            e_ref = ((ref16 - want).abs() * valid).mean().item()
            mxw = want.abs().max().item()
            record_parity(f"full-width decoder, random norm gains, bf16 folded (HF bf16 eager: {e_ref:.4g})", dtype,
                          ((got - want).abs() * valid).max().item(), e_hip, mxw, "mean <= 1.5x HF bf16-eager mean + 1e-3")
            record_parity("full-width decoder, random norm gains, bf16 fold_norms=False", dtype,
                          ((got_unf - want).abs() * valid).max().item(), e_unf, mxw, "mean <= 1.5x HF bf16-eager mean + 1e-3")
            assert e_hip <= 1.5 * e_ref + 1e-3, (e_hip, e_unf, e_ref)
            assert e_unf <= 1.5 * e_ref + 1e-3, (e_hip, e_unf, e_ref)
            assert e_hip <= 1.25 * e_unf + 1e-3, (e_hip, e_unf, e_ref)


@pytest.mark.parametrize("px", [384, 336])
def test_full_width_siglip_tower_vs_transformers(px):
    """SigLIP-so400m/14 at its real width (1152, 16 heads x 72, MLP 4304; 2 of the 27 layers) against transformers'
    SiglipVisionModel on the CPU: 384 px (729 patches, what the reference runs) and 336 px - BASELINE's metric resolution,
    576 patches with the bicubic position-embedding interpolation of HF:siglip/modeling_siglip.py (interpolate_pos_encoding),
    which the reference itself never exercises.  fp32: 1e-5 of max|ref|; bf16: <= 1.5x transformers' own bf16-eager error."""
    from transformers import SiglipVisionConfig, SiglipVisionModel
    from aki_amd.siglip import SiglipVisionTransformer
    torch.manual_seed(0)
    cfg = SiglipVisionConfig(hidden_size=1152, intermediate_size=4304, num_hidden_layers=2, num_attention_heads=16, image_size=384,
                             patch_size=14, attn_implementation="eager")
    hf = SiglipVisionModel(cfg).eval()
    from conftest import randomize_norms_and_biases, record_parity
    # LayerNorm gains 1 + 0.2 N, LayerNorm and linear biases 0.1 N: the folded `W diag(gamma)`, `W beta + b` and `mean * c` terms all live
    assert randomize_norms_and_biases(hf, seed=22) >= 2 * 10 + 3
    x = torch.rand(2, 3, px, px, generator=torch.Generator().manual_seed(1)) * 2 - 1
    with torch.no_grad():
        want = hf(pixel_values=x, interpolate_pos_encoding=(px != 384)).last_hidden_state
        assert want.shape[1] == (px // 14) ** 2
        vt = SiglipVisionTransformer(cfg)
        sd = {k[len("vision_model."):] if k.startswith("vision_model.") else k: v for k, v in hf.state_dict().items()}
        missing = vt.load_state_dict(sd, strict=False)       # the pooling head is not used by AKI
        assert not missing.missing_keys
        got32 = vt.to(DEV).eval()(x.to(DEV), interpolate_pos_encoding=(px != 384)).last_hidden_state.cpu()
        err = (got32 - want).abs().max().item()
        record_parity(f"full-width SigLIP tower at {px} px, fp32 vs transformers", torch.float32, err, (got32 - want).abs().mean().item(),
                      want.abs().max().item(), "1e-5*max(1,max|ref|)")
        assert err <= 1e-5 * max(1.0, want.abs().max().item()), err
        got16 = vt.to(torch.bfloat16)(x.to(DEV).to(torch.bfloat16), interpolate_pos_encoding=(px != 384)).last_hidden_state.float().cpu()
        ref16 = hf.to(torch.bfloat16)(pixel_values=x.to(torch.bfloat16), interpolate_pos_encoding=(px != 384)).last_hidden_state.float()
        vt.encoder.fold_norms = False
        got16u = vt(x.to(DEV).to(torch.bfloat16), interpolate_pos_encoding=(px != 384)).last_hidden_state.float().cpu()
        assert not torch.equal(got16, got16u)
        e_hip, e_unf, e_ref = (got16 - want).abs().mean().item(), (got16u - want).abs().mean().item(), (ref16 - want).abs().mean().item()
        mxw = want.abs().max().item()
        record_parity(f"full-width SigLIP tower at {px} px, random LN gains/biases, bf16 folded (HF bf16 eager: {e_ref:.4g})", torch.bfloat16,
                      (got16 - want).abs().max().item(), e_hip, mxw, "mean <= 1.5x HF bf16-eager mean + 1e-3")
        record_parity(f"full-width SigLIP tower at {px} px, random LN gains/biases, bf16 fold_norms=False", torch.bfloat16,
                      (got16u - want).abs().max().item(), e_unf, mxw, "mean <= 1.5x HF bf16-eager mean + 1e-3")
        assert e_hip <= 1.5 * e_ref + 1e-3, (e_hip, e_unf, e_ref)
        assert e_unf <= 1.5 * e_ref + 1e-3, (e_hip, e_unf, e_ref)
        assert e_hip <= 1.25 * e_unf + 1e-3, (e_hip, e_unf, e_ref)
