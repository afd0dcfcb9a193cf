"""One-sample prefill: the four decoder GEMMs at the row counts of the reference's real callers - M = 655 (one image +
512-token prompt; local_demo.py:75-87, eval_cv_bench/eval.py:92-104) and M = 207 (BASELINE configs[0]) - AS THE MODEL LAUNCHES THEM (folded
RMSNorm on qkv / gate_up, residual + row statistics on o_proj / down, HF:phi3/modeling_phi3.py:287-328) against the numpy oracle, and
bit-identical from launch to launch: at these sizes the library splits K over several workgroups whose f32 partial sums are folded in slice
order by whichever of them arrives last (gemm_bf16.hip, plan_small_m) - the result may not depend on the arrival order."""
import numpy as np
import pytest
import torch

from golden import gen
import aki_oracle as O
from test_kernels_gpu import DEV, check, n, rnd, t, _ops

pytestmark = pytest.mark.gpu
BF = torch.bfloat16
D, F, H = 3072, 8192, 32


def _rstd(y, eps):
    y = y.astype(np.float64)
    return 1.0 / np.sqrt((y * y).mean(-1) + eps)


@pytest.mark.parametrize("M", [207, 655])
@pytest.mark.parametrize("which", ["o_proj", "down"])
def test_prefill_residual_gemms_with_statistics(M, which):
    """h_out = h + x W^T leaving 1/rms(h_out): o_proj (K 3072) and down (K 8192) at N = 3072.  The planner splits K here (the workspace
    query says so - otherwise this test would exercise nothing new); twenty launches give one bit pattern."""
    ops = _ops()
    from aki_amd import _lib
    K = D if which == "o_proj" else F
    assert _lib.load().aki_linear_splitk_workspace_bytes(M, D, K) > 0, "the small-M planner no longer splits K for this shape"
    rng = gen.rng_for(f"prefill{which}{M}")
    x = rng.standard_normal((M, K), dtype=np.float32) * 0.7
    w = rng.standard_normal((D, K), dtype=np.float32) * 0.03
    r = rng.standard_normal((M, D), dtype=np.float32) * 1.5
    xd, wd, rd = t(x, BF), t(w, BF), t(r, BF)
    want = rnd(x, BF) @ rnd(w, BF).T + rnd(r, BF)
    st = ops.new_stats(M, DEV)
    y = ops.linear(xd, wd, residual=rd, stats_out=st, stats_eps=1e-5)
    check(n(y), want, BF, f"{which} M={M}: x W^T + residual")
    np.testing.assert_allclose(n(st.rstd), _rstd(n(y), 1e-5), rtol=3e-5, err_msg="row statistics of the stored output")
    first, first_st = y.clone(), st.rstd.clone()
    for rep in range(20):
        st2 = ops.new_stats(M, DEV)
        y2 = ops.linear(xd, wd, residual=rd, stats_out=st2, stats_eps=1e-5)
        assert torch.equal(y2, first) and torch.equal(st2.rstd, first_st), f"launch {rep + 2} differs from the first: the split-K fold is order-dependent"
    # without the statistics and without a residual: same products, same split
    y3 = ops.linear(xd, wd)
    check(n(y3), rnd(x, BF) @ rnd(w, BF).T, BF, f"{which} M={M}: plain")


@pytest.mark.parametrize("M", [207, 655])
def test_prefill_gate_up_folded_swiglu(M):
    ops = _ops()
    rng = gen.rng_for(f"prefillgu{M}")
    x = rng.standard_normal((M, D), dtype=np.float32) * rng.uniform(0.3, 4.0, (M, 1)).astype(np.float32)
    g = 1.0 + 0.3 * rng.standard_normal((D,), dtype=np.float32)
    w = rng.standard_normal((2 * F, D), dtype=np.float32) * 0.03
    xd = t(x, BF)
    st = ops.row_stats(xd, 1e-5)
    wf = ops.fold_gain(t(w, BF), t(g, BF))
    y = ops.linear(xd, wf, act=ops.ACT_SWIGLU, row_scale=st.rstd)
    up = O.rms_norm(rnd(x, BF), rnd(g, BF), 1e-5) @ rnd(w, BF).T
    check(n(y), up[:, F:] * O.silu(up[:, :F]), BF, f"gate_up M={M}: folded RMSNorm + SwiGLU", scale_atol=2.0)
    assert torch.equal(y, ops.linear(xd, wf, act=ops.ACT_SWIGLU, row_scale=st.rstd))


@pytest.mark.parametrize("M", [207, 655])
def test_prefill_qkv_rope_folded_into_a_kv_cache(M):
    ops = _ops()
    rng = gen.rng_for(f"prefillqkv{M}")
    x = rng.standard_normal((1, M, D), dtype=np.float32) * rng.uniform(0.3, 4.0, (1, M, 1)).astype(np.float32)
    g = 1.0 + 0.3 * rng.standard_normal((D,), dtype=np.float32)
    w = rng.standard_normal((3 * D, D), dtype=np.float32) * 0.03
    cap = M + 40
    cos, sin = O.rope_cos_sin(np.arange(cap)[None], 96)
    cosd, sind = torch.from_numpy(cos[0]).to(DEV), torch.from_numpy(sin[0]).to(DEV)
    xd = t(x, BF)
    st = ops.row_stats(xd, 1e-5)
    wf = ops.fold_gain(t(w, BF), t(g, BF))
    kc = torch.full((1, H, cap, 96), 7.0, dtype=BF, device=DEV)
    vc = torch.full((1, H, cap, 96), 7.0, dtype=BF, device=DEV)
    q, k, v = ops.qkv_rope(xd, wf, cosd, sind, H, k_out=kc, v_out=vc, row_scale=st.rstd)
    qkv = (O.rms_norm(rnd(x, BF), rnd(g, BF), 1e-5).reshape(M, D) @ rnd(w, BF).T).reshape(1, M, 3 * D)
    hd = lambda a: a.reshape(1, M, H, 96).transpose(0, 2, 1, 3)
    qw, kw = O.apply_rope(hd(qkv[..., :D]), hd(qkv[..., D:2 * D]), cos[:, :M], sin[:, :M])
    check(n(q), qw, BF, f"q M={M}", scale_atol=2.0)
    check(n(kc[:, :, :M]), kw, BF, f"k in the cache M={M}", scale_atol=2.0)
    check(n(vc[:, :, :M]), hd(qkv[..., 2 * D:]), BF, f"v in the cache M={M}", scale_atol=2.0)
    assert bool((kc[:, :, M:] == 7.0).all()) and bool((vc[:, :, M:] == 7.0).all()), "rows beyond the prompt were written"
    q2, _, _ = ops.qkv_rope(xd, wf, cosd, sind, H, k_out=kc.clone(), v_out=vc.clone(), row_scale=st.rstd)
    assert torch.equal(q, q2)


def test_generate_first_token_uses_the_last_row_head_and_matches_the_full_logits():
    """`generate`'s prefill runs the head on each sample's last valid token only (Phi3ForCausalLM.forward(last_token_logits=True): a
    weight-streaming GEMV with the final norm inside instead of an L-row GEMM against the 197 MB head).  Same numbers up to bf16 rounding
    as the row of the full logits tensor, for a right-padded batch, and the same arg-max."""
    from aki_amd.factory import build_aki
    from aki_amd.phi3 import make_phi3_config
    from aki_amd.siglip import make_siglip_config
    m = build_aki(lm_config=make_phi3_config(num_hidden_layers=2), vis_config=make_siglip_config(num_hidden_layers=1, image_size=224),
                  dtype=BF, device=DEV, seed=9).eval()
    g = torch.Generator().manual_seed(2)
    B, T = 3, 48
    ids = torch.randint(3, 32000, (B, T), generator=g)
    am = torch.ones(B, T, dtype=torch.long)
    for b, nreal in enumerate((48, 40, 31)):
        ids[b, 0], ids[b, 5] = 1, m.media_token_id
        ids[b, nreal:], am[b, nreal:] = 32000, 0
    vx = ((torch.rand((B, 1, 1, 3, 224, 224), generator=g) - 0.5) / 0.5).to(DEV, BF)
    ids, am = ids.to(DEV), am.to(DEV)
    with torch.no_grad():
        plan = m._start_splice_plan(ids)
        vt = m.vision_tokenizer(m._encode_vision_x(vision_x=vx))
        prep = m._prepare_inputs_for_forward(vision_tokens=vt, lang_x=ids, attention_mask=am, padding_side="right", splice_plan=plan)
        full = m.lang_model(inputs_embeds=prep["inputs_embeds"], attention_mask=prep["attention_mask"], use_cache=True)
        last = m.lang_model(inputs_embeds=prep["inputs_embeds"], attention_mask=prep["attention_mask"], use_cache=True, last_token_logits=True)
    assert last.logits.shape == (B, 1, full.logits.shape[-1])
    rows = (full.past_key_values.cache_len.long() - 1)
    want = full.logits[torch.arange(B, device=DEV), rows].float()
    got = last.logits[:, 0].float()
    check(n(got), n(want), BF, "last-row head vs the same row of the full logits", scale_atol=2.0)
    assert torch.equal(got.argmax(-1), want.argmax(-1))
    assert torch.equal(last.past_key_values.cache_len, full.past_key_values.cache_len)
