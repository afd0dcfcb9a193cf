"""BASELINE configs[4]: fp8 (OCP e4m3) projections.  No reference behaviour exists for this configuration (parity unpinned
by the reference); the checks are (1) the quantiser against torch's float8_e4m3fn cast, bit for bit, (2) the fp8 MFMA GEMM
against exact f32 arithmetic on the SAME quantised bytes, (3) the fused MMA op and a small Phi-3 stack against the bf16
path within the error that per-row e4m3 quantisation of operands implies."""
import numpy as np
import pytest
import torch

from golden import gen
import aki_oracle as O

pytestmark = pytest.mark.gpu
DEV, BF = "cuda", torch.bfloat16


def rt(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(BF).to(DEV)


def deq(q, s):
    return q.view(torch.float8_e4m3fn).float() * s[:, None]


@pytest.mark.parametrize("rows,cols,rms", [(37, 3072, False), (300, 8192, False), (64, 3072, True), (5, 192, True)])
def test_quantiser_matches_torch_e4m3_cast(rows, cols, rms):
    from aki_amd import ops
    x = rt(rows, cols, seed=1, scale=3.0)
    x[0, :5] = torch.tensor([448.0, -448.0, 1e-3, 0.0, 17.0], dtype=BF)
    w = (1 + 0.1 * torch.randn(cols)).to(BF).to(DEV)
    q, s = ops.quant_rows_fp8(x, w if rms else None, 1e-5)
    y = ops.rmsnorm(x, w, 1e-5).float() if rms else x.float()
    s_ref = y.abs().amax(-1).clamp(min=1e-12) * (1.0 / 448.0)
    assert torch.allclose(s, s_ref, rtol=1e-6, atol=0)
    t = (y * (1.0 / s_ref)[:, None]).clamp(-448, 448).to(torch.float8_e4m3fn).view(torch.uint8)
    same = (q == t).float().mean().item()
    assert same > 0.9999, f"{(1 - same) * q.numel():.0f} of {q.numel()} bytes differ from torch's RNE e4m3 cast"
    assert (deq(q, s) - y).abs().max().item() <= 0.0625 * y.abs().amax().item() + 1e-6   # e4m3: 3 mantissa bits


@pytest.mark.parametrize("M,N,K,act", [(5240, 3072, 3072, "none"), (700, 1024, 512, "swiglu"), (333, 32016, 3072, "bias"), (64, 3072, 8192, "res"),
                                       # BASELINE configs[4]'s own shapes: batch 16 x 655 = 10 480 token rows (gate_up + SwiGLU, down + residual)
                                       (10480, 16384, 3072, "swiglu"), (10480, 3072, 8192, "res"), (10480, 9216, 3072, "none")])
def test_fp8_gemm_exact_on_quantised_operands(M, N, K, act):
    from aki_amd import ops
    x, w = rt(M, K, seed=2), rt(N, K, seed=3, scale=0.05)
    xq, xs = ops.quant_rows_fp8(x)
    wq, ws = ops.quant_rows_fp8(w)
    ref = deq(xq, xs) @ deq(wq, ws).t()
    kw = {}
    if act == "swiglu":
        g, u = ref[:, : N // 2], ref[:, N // 2:]
        ref = u * torch.nn.functional.silu(g)
        kw["act"] = ops.ACT_SWIGLU
    if act == "bias":
        b = rt(N, seed=4, scale=0.1)
        ref = ref + b.float()
        kw["bias"] = b
    if act == "res":
        r = rt(M, N, seed=5)
        ref = ref + r.float()
        kw["residual"] = r
    y = ops.linear_fp8(xq, xs, wq, ws, **kw).float()
    err = (y - ref).abs()
    tol = 2 ** -8 * ref.abs() + 1e-3 * ref.abs().max()
    assert bool((err <= tol).all()), f"max err {err.max().item():.4g} at |ref| max {ref.abs().max().item():.4g}"


def test_fp8_fused_mma_op_vs_bf16_path():
    from aki_amd import ops
    B, L, H, Dh = 2, 200, 4, 96
    d = H * Dh
    x, w = rt(B, L, d, seed=6), rt(3 * d, d, seed=7, scale=0.05)
    cos, sin = O.rope_cos_sin(np.arange(L)[None], Dh)
    tc, ts = torch.from_numpy(cos[0]).to(DEV), torch.from_numpy(sin[0]).to(DEV)
    am = np.ones((B, L), dtype=bool)
    am[1, 180:] = False
    table = ops.MaskTable.from_host([[(6, 150, 150, 190)], [(0, 0, 0, 0)]], am, [L, L], DEV)
    ref = ops.mma_attn(x, w, tc, ts, table, H).float()
    xq, xs = ops.quant_rows_fp8(x)
    wq, ws = ops.quant_rows_fp8(w)
    # the bf16 kernel on the DEQUANTISED operands isolates the GEMM precision from the quantisation error
    xd, wd = deq(xq, xs).to(BF).view(B, L, d), deq(wq, ws).to(BF)
    mid = ops.mma_attn(xd, wd, tc, ts, table, H).float()
    got = ops.mma_attn_fp8(xq, xs, wq, ws, tc, ts, table, B, H).float()
    valid = torch.from_numpy(am).to(DEV)[..., None]
    e_kernel = ((got - mid).abs() * valid).max().item()
    e_quant = ((got - ref).abs() * valid).max().item()
    assert e_kernel <= 2e-2 * mid.abs().max().item(), e_kernel          # same math, bf16-rounding of dequantised operands apart
    assert e_quant <= 0.15 * ref.abs().max().item(), e_quant            # what e4m3 operands cost


def test_fp8_phi3_stack_tracks_bf16():
    from aki_amd import ops
    from aki_amd.phi3 import Phi3ForCausalLM, make_phi3_config
    torch.manual_seed(0)
    cfg = make_phi3_config(vocab_size=1024, hidden_size=384, intermediate_size=1024, num_hidden_layers=3, num_attention_heads=4,
                           num_key_value_heads=4, pad_token_id=0)
    lm = Phi3ForCausalLM(cfg)
    for p in lm.parameters():
        if p.dim() > 1:
            p.data.normal_(0, 0.05)
    lm = lm.to(DEV).to(BF).eval()
    B, L = 2, 130
    x = rt(B, L, 384, seed=8, scale=0.5)
    am = np.ones((B, L), dtype=bool)
    table = ops.MaskTable.from_host([[(2, 60, 60, 120)]] * B, am, [L] * B, DEV)
    with torch.no_grad():
        ref = lm(inputs_embeds=x, attention_mask=table).logits.float()
        lm.enable_fp8()
        got = lm(inputs_embeds=x, attention_mask=table).logits.float()
        lm.enable_fp8(False)
        back = lm(inputs_embeds=x, attention_mask=table).logits.float()
    assert torch.equal(back, ref)                                       # switching off restores the bf16 path exactly
    # e4m3 keeps 3 mantissa bits: ~3.5 % relative error per GEMM with random operands, 13 GEMMs deep here -> ~sqrt(13) * 3.5 %
    rel = ((got - ref).norm() / ref.norm()).item()
    assert rel < 0.2, f"fp8 logits differ from bf16 by relative L2 {rel:.3f}"
    cosine = torch.nn.functional.cosine_similarity(got.flatten(), ref.flatten(), dim=0).item()
    assert cosine > 0.98, cosine


@pytest.mark.parametrize("N,K,mode", [(3072, 3072, "plain"), (9216, 3072, "rms"), (16384, 3072, "swiglu_rms"), (3072, 8192, "res"), (32016, 3072, "bias")])
def test_w8a16_gemv_on_quantised_weights(N, K, mode):
    """Weight-only fp8 GEMV (single-sequence decode in the fp8 configuration): exact arithmetic on the quantised weights."""
    from aki_amd import ops
    x, w = rt(1, K, seed=30, scale=2.0), rt(N, K, seed=31, scale=0.05)
    wq, ws = ops.quant_rows_fp8(w)
    g = (1 + 0.1 * torch.randn(K)).to(BF).to(DEV)
    kw, xin = {}, x
    if "rms" in mode:
        kw.update(rms_weight=g, eps=1e-5)
        xin = ops.rmsnorm(x, g, 1e-5)
    ref = xin.float() @ deq(wq, ws).t()
    if "swiglu" in mode:
        kw["act"] = ops.ACT_SWIGLU
        ref = ref[:, N // 2:] * torch.nn.functional.silu(ref[:, : N // 2])
    if mode == "res":
        r = rt(1, N, seed=32)
        kw["residual"] = r
        ref = ref + r.float()
    if mode == "bias":
        b = rt(N, seed=33, scale=0.1)
        kw["bias"] = b
        ref = ref + b.float()
    y = ops.linear_w8(x, wq, ws, **kw).float()
    err = (y - ref).abs()
    assert bool((err <= 2 ** -7 * ref.abs() + 2e-3 * ref.abs().max()).all()), err.max().item()


def test_fp8_prefill_into_cache_and_w8_decode():
    """fp8 configuration end to end on a small Phi-3 stack: the prefill that writes the KV cache equals the cache-less fp8
    forward, and w8a16 decode steps stay close to a full fp8 forward over the extended sequence (the decode side keeps bf16
    activations, so it is the MORE accurate of the two)."""
    from aki_amd import ops
    from aki_amd.phi3 import Phi3ForCausalLM, make_phi3_config
    torch.manual_seed(0)
    cfg = make_phi3_config(vocab_size=1024, hidden_size=384, intermediate_size=1024, num_hidden_layers=3, num_attention_heads=4,
                           num_key_value_heads=4, pad_token_id=0)
    lm = Phi3ForCausalLM(cfg)
    for p in lm.parameters():
        if p.dim() > 1:
            p.data.normal_(0, 0.05)
    lm = lm.to(DEV).to(BF).eval().enable_fp8()
    L0 = 70
    ids = torch.randint(1, 1000, (1, L0), generator=torch.Generator().manual_seed(3)).to(DEV)
    with torch.no_grad():
        emb = lm.get_input_embeddings()(ids)
        table = ops.MaskTable.from_host([[(2, 30, 30, 60)]], np.ones((1, L0), dtype=bool), [L0], DEV)
        plain = lm(inputs_embeds=emb, attention_mask=table).logits
        out = lm(inputs_embeds=emb, attention_mask=table, use_cache=True, cache_capacity=L0 + 8)
        assert torch.equal(out.logits, plain)                      # same kernels; K/V merely land in the cache
        cache = out.past_key_values
        nxt = out.logits[0, -1].float().argmax()[None]
        seq = ids
        for step in range(3):
            dec = lm.decode_step(input_ids=nxt, past_key_values=cache)[0].float()
            seq = torch.cat([seq, nxt[None]], 1)
            Ls = seq.shape[1]
            tb = ops.MaskTable.from_host([[(2, 30, 30, 60)]], np.ones((1, Ls), dtype=bool), [Ls], DEV)
            full = lm(inputs_embeds=lm.get_input_embeddings()(seq), attention_mask=tb).logits[0, -1].float()
            rel = ((dec - full).norm() / full.norm()).item()
            assert rel < 0.15, (step, rel)
            nxt = dec.argmax()[None]
