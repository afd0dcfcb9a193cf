"""Pin the numpy oracle (oracle/aki_oracle.py) to golden vectors produced by the reference itself
(tests/golden/make_golden.py).  CPU only."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR, load_golden
from golden import gen
import aki_oracle as O


def _shapes(g):
    return [(k, tuple(s)) for k, s in json.loads(str(g["shapes"]))]


def test_mask_bit_exact_vs_reference():
    g = load_golden("mask_cases.npz")
    n_cases = int(g["n_cases"])
    assert n_cases >= 20
    cases = gen.mask_cases()
    assert len(cases) == n_cases
    for i, (am, s, t, e) in enumerate(cases):
        assert np.array_equal(g[f"am_{i}"].astype(np.int64), am)
        assert list(g[f"args_{i}"]) == [s, t, e]
        n = len(am)
        want = gen.unpack_mask_bits(g[f"bits_{i}"], (1, n, n))
        got = O.make_modality_mutual_mask(am, s, t, e)
        assert got.dtype == np.int64 and np.array_equal(got, want), f"dense restatement, case {i}"
        got2 = O.mask_from_spans(am, [O.clamp_span(n, s, t, e)])
        assert np.array_equal(got2, want), f"closed form / spans, case {i}"


def test_decoupled_embedding_and_linear():
    g = load_golden("decoupled.npz")
    V = int(g["V"])
    e = O.decoupled_embedding(g["ids"], g["W"], g["Wadd"], V - 1)
    assert np.array_equal(e, g["emb"])
    y = O.decoupled_linear(g["x"], g["W"], None, g["Wl"], None, V - 1)
    assert y.shape == g["y"].shape
    np.testing.assert_allclose(y, g["y"], atol=2e-6, rtol=1e-6)


def test_longrope_tables():
    g = load_golden("rope_longrope.npz")
    sc = float(g["attention_scaling"])
    assert abs(sc - O.longrope_attention_scaling(131072, 4096)) < 1e-6
    # fp32 pow() differs by an ulp between numpy and torch; the angle error grows with the position
    for pos, fac, cw, sw in ((g["pos_s"], g["short"], g["cos_s"], g["sin_s"]), (g["pos_l"], g["long"], g["cos_l"], g["sin_l"])):
        c, s = O.rope_cos_sin(pos, 96, 10000.0, fac, sc)
        tol = (2e-6 + 4e-7 * pos.astype(np.float64))[..., None]
        assert np.all(np.abs(c - cw) <= tol) and np.all(np.abs(s - sw) <= tol)


@pytest.mark.parametrize("tag", ["small", "mid"])
def test_phi3_attention_block(tag):
    g = load_golden(f"attn_block_{tag}.npz")
    p = gen.fill_params(_shapes(g), 31)
    B, L = g["am"].shape
    d = p["qkv_proj.weight"].shape[1]
    H = d // 96
    x = gen.rng_for("attn_block_" + tag).standard_normal((B, L, d), dtype=np.float32)
    m4 = gen.unpack_mask_bits(g["mask_bits"], tuple(g["mask_shape"]))
    # dense mask == span restatement
    rects = [[O.clamp_span(L, *map(int, g["spans"][b]))] for b in range(B)]
    for b in range(B):
        assert np.array_equal(O.mask_from_spans(g["am"][b], rects[b]), m4[b])
    cos, sin = O.rope_cos_sin(np.arange(L)[None], 96)
    tol = (2e-6 + 4e-7 * np.arange(L, dtype=np.float64))[None, :, None]
    assert np.all(np.abs(cos - g["cos"]) <= tol) and np.all(np.abs(sin - g["sin"]) <= tol)
    cos, sin = g["cos"], g["sin"]          # tables are inputs of the attention block; use the reference's
    y = O.phi3_attention(x, p["qkv_proj.weight"], p["o_proj.weight"], cos, sin, O.invert_mask_441(m4), H)
    np.testing.assert_allclose(y, g["y32"], atol=1e-5, rtol=1e-5)
    # span-driven core (what the kernel computes) equals the dense-mask core on every row
    qkv = x @ p["qkv_proj.weight"].T
    hd = lambda t: t.reshape(B, L, H, 96).transpose(0, 2, 1, 3)
    q, k = O.apply_rope(hd(qkv[..., :d]), hd(qkv[..., d:2 * d]), cos, sin)
    v = hd(qkv[..., 2 * d:])
    o_dense = O.mma_attention_core(q, k, v, O.invert_mask_441(m4), 96 ** -0.5)
    o_span = O.mma_attention_core_spans(q, k, v, g["am"], rects, 96 ** -0.5)
    np.testing.assert_allclose(o_span, o_dense, atol=2e-6, rtol=1e-5)
    # bf16 eager emulation tracks the reference's bf16 eager output to bf16 resolution
    xb = O.bf16_round(x)
    pb = {k_: O.bf16_round(v_) for k_, v_ in p.items()}
    y16 = O.phi3_attention(xb, pb["qkv_proj.weight"], pb["o_proj.weight"], cos, sin,
                           O.invert_mask_441(m4, O.BF16_MIN), H, emulate_bf16=True)
    err = np.abs(y16 - g["y16"])
    scale = np.abs(g["y16"]).max()
    assert err.max() <= 2 ** -6 * scale and err.mean() <= 2e-3 * scale


def test_perceiver_small_and_full():
    g = load_golden("perceiver_small.npz")
    p = gen.fill_params(_shapes(g), 21)
    x = gen.rng_for("perceiver_small").standard_normal((2, 1, 1, 16, 64), dtype=np.float32)
    y = O.perceiver_resampler(x, p)
    np.testing.assert_allclose(y, g["y"], atol=2e-5, rtol=1e-4)
    g = load_golden("perceiver_full.npz")
    p = gen.fill_params(_shapes(g), 22)
    x = gen.rng_for("perceiver_full").standard_normal((1, 1, 1, 729, 1152), dtype=np.float32)
    y = O.perceiver_resampler(x, p)
    np.testing.assert_allclose(y[0, 0, g["rows"]], g["y_rows"], atol=5e-4, rtol=1e-3)
    assert abs(float(np.abs(y.astype(np.float64)).sum()) - float(g["y_abs"])) <= 1e-4 * float(g["y_abs"])


def test_patch_embed_full_dims():
    g = load_golden("patch_embed_full.npz")
    p = gen.fill_params(_shapes(g), 41)
    x = gen.rng_for("patch_embed").random((2, 3, 384, 384), dtype=np.float32) * 2 - 1
    y = O.siglip_patch_embed(x, p["patch_embedding.weight"], p["patch_embedding.bias"], p["position_embedding.weight"])
    assert y.shape == (2, 729, 1152)
    np.testing.assert_allclose(y[:, g["rows"]], g["y_rows"], atol=2e-5, rtol=1e-5)
    assert abs(float(np.abs(y.astype(np.float64)).sum()) - float(g["y_abs"])) <= 1e-5 * float(g["y_abs"])


def tiny_cfg():
    T = gen.TINY
    return dict(vis_layers=T["vis_layers"], vis_heads=T["vis_heads"], lm_layers=T["lm_layers"], lm_heads=T["lm_heads"],
                max_original_id=T["vocab"] - 1, media_token_id=T["media_token_id"], pad_token_id=T["pad_token_id"],
                num_vision_tokens=T["num_vision_tokens"])


def test_tiny_aki_end_to_end():
    g = load_golden("tiny_e2e.npz")
    p = gen.fill_params(_shapes(g), 11)
    rng = gen.rng_for("tiny_batch")
    B = g["lang_x"].shape[0]
    vision_x = rng.standard_normal((B, 1, 1, 3, gen.TINY["image"], gen.TINY["image"]), dtype=np.float32)
    out = O.aki_forward(p, tiny_cfg(), vision_x, g["lang_x"], g["attention_mask"], g["labels"])
    prep = out["prep"]
    np.testing.assert_allclose(prep["inputs_embeds"], g["inputs_embeds"], atol=2e-5, rtol=1e-4)
    assert np.array_equal(prep["attention_mask"], gen.unpack_mask_bits(g["mask_bits"], tuple(g["mask_shape"])))
    assert np.array_equal(prep["labels"], g["new_labels"])
    np.testing.assert_allclose(out["logits"][:, :, g["logit_cols"]], g["logits"], atol=2e-4, rtol=1e-3)
    assert abs(out["loss"] - float(g["loss"])) < 1e-4
    # pad rows of inputs_embeds carry the scalar pad_token_id (src/vlm.py:584-588)
    L0 = prep["lengths"][1]
    assert np.all(prep["inputs_embeds"][1, L0:] == float(gen.TINY["pad_token_id"]))


def test_multi_image_raises_like_reference():
    T = gen.TINY
    lang_x = np.array([[1, T["media_token_id"], 5, T["media_token_id"], 6, 32001, 7]])
    emb = np.zeros((1, 7, 4), dtype=np.float32)
    vt = np.zeros((1, 2, 8, 4), dtype=np.float32)
    with pytest.raises(RuntimeError):
        O.prepare_inputs_for_forward(vt, lang_x, np.ones_like(lang_x), None, emb, T["media_token_id"], T["pad_token_id"], 8)


def test_fixture_files_are_data_only():
    for f in os.listdir(GOLDEN_DIR):
        if f.endswith(".npz"):
            assert os.path.getsize(os.path.join(GOLDEN_DIR, f)) < 2 * 1024 * 1024


# ---- torch restatement (oracle/aki_torch.py): forward and autograd gradients pinned to the reference -------------
def _tiny_torch():
    import torch
    import aki_torch as OT
    g = load_golden("tiny_e2e.npz")
    p = {k: torch.from_numpy(v.copy()) for k, v in gen.fill_params(_shapes(g), 11).items()}
    rng = gen.rng_for("tiny_batch")
    B = g["lang_x"].shape[0]
    vision_x = torch.from_numpy(rng.standard_normal((B, 1, 1, 3, gen.TINY["image"], gen.TINY["image"]), dtype=np.float32))
    return OT, g, p, vision_x


def test_torch_oracle_forward_matches_reference():
    import torch
    OT, g, p, vision_x = _tiny_torch()
    with torch.no_grad():
        out = OT.aki_forward(p, tiny_cfg(), vision_x, torch.from_numpy(g["lang_x"]), torch.from_numpy(g["attention_mask"]),
                             torch.from_numpy(g["labels"]))
    np.testing.assert_allclose(out["prep"]["inputs_embeds"].numpy(), g["inputs_embeds"], atol=2e-5, rtol=1e-4)
    assert np.array_equal(out["prep"]["labels"].numpy(), g["new_labels"])
    np.testing.assert_allclose(out["logits"][:, :, torch.from_numpy(g["logit_cols"])].numpy(), g["logits"], atol=2e-4, rtol=1e-3)
    assert abs(float(out["loss"]) - float(g["loss"])) < 1e-4


def test_torch_oracle_gradients_match_reference_backward():
    """autograd over the restatement == the reference's loss.backward() (tiny_grads.npz) for every trainable parameter."""
    import json, torch
    OT, g, p, vision_x = _tiny_torch()
    gg = load_golden("tiny_grads.npz")
    names = json.loads(str(gg["names"]))
    for n_ in names:
        p[n_].requires_grad_(True)
    out = OT.aki_forward(p, tiny_cfg(), vision_x, torch.from_numpy(g["lang_x"]), torch.from_numpy(g["attention_mask"]),
                         torch.from_numpy(g["labels"]))
    out["loss"].backward()
    assert abs(float(out["loss"]) - float(gg["loss"])) < 1e-5
    assert not any(k.startswith("vision_encoder.") for k in names)          # frozen tower (src/aki.py:52-57)
    for i, n_ in enumerate(names):
        gr = p[n_].grad
        gr = torch.zeros_like(p[n_]) if gr is None else gr
        idx = gen.grad_sample_idx(gr.numel())
        got = gr.flatten()[torch.from_numpy(idx)].numpy()
        want = gg["samples"][i][: len(idx)]
        scale = max(float(gg["abss"][i]) / gr.numel(), 1e-8)
        np.testing.assert_allclose(got, want, atol=2e-3 * scale + 1e-9, rtol=2e-3, err_msg=n_)
        assert abs(float(gr.double().abs().sum()) - float(gg["abss"][i])) <= 1e-3 * float(gg["abss"][i]) + 1e-9, n_


def test_mask_to_table_round_trips_every_reference_mask():
    """oracle.mask_to_table (the restatement the HIP converter aki_mma_mask_to_table is checked against) inverts the
    reference's `_make_modality_mutual_mask` on all golden cases: table -> dense gives the reference's mask back."""
    g = load_golden("mask_cases.npz")
    for i, (am, s, t, e) in enumerate(gen.mask_cases()):
        n = len(am)
        dense = gen.unpack_mask_bits(g[f"bits_{i}"], (1, n, n))[0]
        out = O.mask_to_table(dense, max_rects=1)
        assert out is not None, i
        rects, valid, seq_len = out
        back = O.mask_from_spans(valid.astype(np.int64), rects)[0].copy()
        back[seq_len:] = 0
        assert np.array_equal(back, dense), i
        want = O.clamp_span(n, s, t, e)
        if rects:      # the rectangle covers the reference's slice assignment up to columns nobody can see / the diagonal
            (r0, r1, c0, c1), = rects
            assert want[0] <= r0 and r1 <= want[1] and c1 <= want[3], (i, rects, want)
    # multi-image (build-defined) and a rectangle dipping below the diagonal
    am = np.ones(60, dtype=np.int64)
    for rr in ([(2, 10, 10, 40), (12, 20, 20, 40)], [(5, 30, 12, 50)]):
        dense = O.mask_from_spans(am, rr)[0]
        rects, valid, seq_len = O.mask_to_table(dense)
        assert np.array_equal(O.mask_from_spans(valid.astype(np.int64), rects)[0], dense) and seq_len == 60
    # a bidirectional mask is ONE rectangle whose lower part the causal triangle already covers
    rects, valid, seq_len = O.mask_to_table(np.ones((40, 40), dtype=np.int64), max_rects=1)
    assert rects == [(0, 39, 1, 40)] and valid.all() and seq_len == 40
    # lookahead windows of different widths: one row group per row -> not representable
    stair = np.tril(np.ones((40, 40), dtype=np.int64))
    for r in range(0, 20, 2):
        stair[r, r + 1:r + 3 + r] = 1
    assert O.mask_to_table(stair, max_rects=8) is None


def test_sft_collate_restatement_vs_reference():
    """oracle.sft_batch_collate_pad against the reference's `batch_collate_pad` (train/sft_data_utils/loader_utils.py:53-91) on
    seeded ragged batches: both padding modes, both sides, truncation, tensors as samples."""
    g = load_golden("sft_collate.npz")
    cases = gen.sft_cases()
    assert int(g["n_cases"]) == len(cases) >= 6
    for i, (batch, padding, side, pad_id, max_length) in enumerate(cases):
        got = O.sft_batch_collate_pad(batch, padding, side, pad_id, max_length)
        for k in ("input_ids", "labels", "attention_mask"):
            assert got[k].dtype == np.int64 and np.array_equal(got[k], g[f"{k}_{i}"]), (i, k)
        longest = max(len(s["input_ids"]) for s in batch)
        assert got["input_ids"].shape[1] == (longest if padding == "longest" else max_length + 1)


def test_torch_oracle_greedy_continuation_matches_the_reference():
    """f1 (generate): the oracle's repeated full forwards reproduce the continuation the REFERENCE produced
    (tests/golden/tiny_generate.npz, made by make_golden.py::g_tiny_generate from the imported reference)."""
    import torch
    OT, g, p, vision_x = _tiny_torch()
    gg = load_golden("tiny_generate.npz")
    cols = torch.from_numpy(gg["logit_cols"])
    with torch.no_grad():
        for b in range(g["lang_x"].shape[0]):
            ids = torch.from_numpy(g["lang_x"][b, : int(g["attention_mask"][b].sum())])[None]
            for t in range(gg["tokens"].shape[1]):
                logits = OT.aki_forward(p, tiny_cfg(), vision_x[b:b + 1], ids, torch.ones_like(ids))["logits"][0, -1]
                np.testing.assert_allclose(logits[cols].numpy(), gg["logits"][b, t], atol=2e-4, rtol=1e-3)
                assert int(logits.argmax()) == int(gg["tokens"][b, t]), (b, t)
                ids = torch.cat([ids, torch.tensor([[int(gg["tokens"][b, t])]])], dim=1)


def test_fake_quant_e4m3_oracle_is_the_plain_oracle_plus_format_noise():
    """BASELINE configs[4] (build-defined; the reference has no fp8): the fake-quant mode of the torch oracle.  (1) its quantiser
    returns e4m3-representable values whose per-row scale maps the row's amax to 448; (2) with quantisation switched off (fp8=None)
    it IS the pinned plain oracle; (3) on the tiny model it tracks the reference's logits within the e4m3 format's noise and
    moves towards them as projections return to the model dtype - the property the full-depth GPU test leans on."""
    import torch
    OT, g, p, vision_x = _tiny_torch()
    x = torch.randn(7, 96, generator=torch.Generator().manual_seed(0)) * 3
    q, s = OT.quant_rows_e4m3(x)
    assert torch.equal(q, q.to(torch.float8_e4m3fn).float()) and float(q.abs().max()) == 448.0
    assert torch.allclose((q * s).abs().amax(-1), x.abs().amax(-1), rtol=1e-6)
    assert float(((q * s) - x).abs().max()) <= 0.0625 * float(x.abs().max())
    ids, am = torch.from_numpy(g["lang_x"]), torch.from_numpy(g["attention_mask"])
    with torch.no_grad():
        plain = OT.aki_forward(p, tiny_cfg(), vision_x, ids, am)["logits"]
        same = OT.aki_forward(p, dict(tiny_cfg(), fp8=None), vision_x, ids, am)["logits"]
        full = OT.aki_forward(p, dict(tiny_cfg(), fp8=dict(head=True, residual_writers=True)), vision_x, ids, am)["logits"]
        part = OT.aki_forward(p, dict(tiny_cfg(), fp8=dict(head=False, residual_writers=False)), vision_x, ids, am)["logits"]
    assert torch.equal(plain, same)
    rel = lambda a: float((a - plain).norm() / plain.norm())
    assert 0.0 < rel(part) < rel(full) < 0.25, (rel(part), rel(full))


def test_multi_image_oracle_extension_reduces_to_the_reference_for_one_image():
    """oracle/aki_torch.py::prepare_inputs_multi_image is build-defined (the reference raises on a second image); with ONE image it must be
    the reference's splice and mask bit for bit (the pinned numpy restatement), for placeholders in front of and behind <|assistant|>, with
    padding, and with no assistant token at all."""
    import aki_torch as OT
    import torch
    g = torch.Generator().manual_seed(3)
    T_, d, Nv = 24, 8, 5
    for img_at, q_at, n_real in ((2, 15, 24), (0, 9, 20), (17, 6, 24), (4, None, 22)):
        lx = torch.randint(3, 100, (1, T_), generator=g)
        lx[0, img_at] = 500
        if q_at is not None:
            lx[0, q_at] = O.ASSISTANT_TOKEN_ID
        am = torch.ones(1, T_, dtype=torch.long)
        am[0, n_real:] = 0
        emb, vt = torch.randn(1, T_, d, generator=g), torch.randn(1, 1, Nv, d, generator=g)
        a = OT.prepare_inputs_for_forward(vt, lx, am, None, emb, 500, 0, Nv)
        b = OT.prepare_inputs_multi_image(vt, lx, am, emb, 500, 0, Nv)
        assert torch.equal(a["inputs_embeds"], b["inputs_embeds"]) and torch.equal(a["attention_mask"], b["attention_mask"]), (img_at, q_at)
        assert np.array_equal(a["mask_1d"], b["mask_1d"])
    # two images: both are spliced, each image's rows see [its end, <|assistant|>]
    lx = torch.randint(3, 100, (1, T_), generator=g)
    lx[0, 1], lx[0, 8], lx[0, 20] = 500, 500, O.ASSISTANT_TOKEN_ID
    emb, vt = torch.randn(1, T_, d, generator=g), torch.randn(1, 2, Nv, d, generator=g)
    b = OT.prepare_inputs_multi_image(vt, lx, torch.ones(1, T_, dtype=torch.long), emb, 500, 0, Nv)
    assert b["inputs_embeds"].shape[1] == T_ + 2 * (Nv - 1) and b["spans"][0] == [(1, 6, 6, 29), (12, 17, 17, 29)]
    assert torch.equal(b["inputs_embeds"][0, 12:17], vt[0, 1]) and torch.equal(b["inputs_embeds"][0, 1:6], vt[0, 0])
    m = b["attention_mask"][0, 0]
    assert int(m[3, 20]) == 1 and int(m[3, 29]) == 0 and int(m[14, 20]) == 1 and int(m[14, 6]) == 1 and int(m[7, 20]) == 0
