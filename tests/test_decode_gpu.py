"""Decode path: GEMV linear for M <= 8, RoPE + KV-cache append, single-query attention, and
AKI.generate() consistency: every decode step must reproduce the logits of a fresh full (MMA prefill) forward over
the prompt extended by the tokens generated so far - the reference's semantics after its prefill (all-ones mask,
src/aki_generation.py:58-62)."""
import numpy as np
import pytest
import torch

from golden import gen
import aki_oracle as O
from test_kernels_gpu import check, n, rnd, t, DEV, DTYPES
from test_model_gpu import build_tiny, batch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("M", [1, 2, 3, 8, 16])
@pytest.mark.parametrize("N,K", [(3072, 3072), (1000, 192), (32016, 3072), (3072, 8192), (9216, 3072)])
def test_gemv_linear_bf16(M, N, K):
    """M = 1: dot-product GEMV; 2 <= M <= 16: skinny MFMA GEMM (every K split: 8 / 4 / 2 / 1 waves per tile)."""
    from aki_amd import ops
    rng = gen.rng_for(f"gemv{M}{N}{K}")
    x = rng.standard_normal((M, K), dtype=np.float32)
    w = rng.standard_normal((N, K), dtype=np.float32) * 0.05
    b = rng.standard_normal((N,), dtype=np.float32) * 0.1
    r = rng.standard_normal((M, N), dtype=np.float32)
    dt = torch.bfloat16
    xr, wr, br, rr = (rnd(a, dt) for a in (x, w, b, r))
    base = xr @ wr.T
    check(n(ops.linear(t(x, dt), t(w, dt))), base, dt, "gemv plain")
    check(n(ops.linear(t(x, dt), t(w, dt), bias=t(b, dt), residual=t(r, dt))), base + br + rr, dt, "gemv bias+residual")
    check(n(ops.linear(t(x, dt), t(w, dt), bias=t(b, dt), act=ops.ACT_GELU_TANH)), O.gelu_tanh((base + br).astype(np.float32)), dt, "gemv gelu")
    if N % 2 == 0:
        up = base
        check(n(ops.linear(t(x, dt), t(w, dt), act=ops.ACT_SWIGLU)), up[:, N // 2:] * O.silu(up[:, :N // 2].astype(np.float32)), dt, "gemv swiglu")


@pytest.mark.parametrize("dtype", DTYPES)
def test_rope_append_and_decode_attention(dtype):
    from aki_amd import ops
    B, H, Dh, cap = 3, 4, 96, 300
    rng = gen.rng_for("decode_attn")
    kc = rng.standard_normal((B, H, cap, Dh), dtype=np.float32)
    vc = rng.standard_normal((B, H, cap, Dh), dtype=np.float32)
    qkv = rng.standard_normal((B, 3 * H * Dh), dtype=np.float32)
    lens = np.array([17, 200, 299 - 1], dtype=np.int32)
    cos, sin = O.rope_cos_sin(np.arange(cap)[None], Dh)
    tk, tv = t(kc, dtype), t(vc, dtype)
    cache_len = torch.from_numpy(lens).to(DEV)
    q = ops.rope_append(t(qkv, dtype), torch.from_numpy(cos[0]).to(DEV), torch.from_numpy(sin[0]).to(DEV), cache_len, cache_len, tk, tv, H)
    qr = rnd(qkv, dtype).reshape(B, 3, H, Dh)
    kcr, vcr = rnd(kc, dtype), rnd(vc, dtype)
    want_q = np.zeros((B, H, Dh), dtype=np.float32)
    for b in range(B):
        c, s = cos[0][lens[b]][None], sin[0][lens[b]][None]
        want_q[b] = qr[b, 0] * c + O.rotate_half(qr[b, 0]) * s
        kcr[b, :, lens[b]] = rnd(qr[b, 1] * c + O.rotate_half(qr[b, 1]) * s, dtype)
        vcr[b, :, lens[b]] = qr[b, 2]
    check(n(q), want_q, dtype, "rotated q")
    check(n(tk), kcr, dtype, "k cache after append")
    assert np.array_equal(n(tv), vcr), "v append must be a plain copy"
    # attention of the new token over [0, len] with a hole in sample 1's prompt mask
    am = np.ones((B, 320), dtype=bool)
    am[1, 5:9] = False
    table = ops.MaskTable.from_host([[(0, 0, 0, 0)]] * B, am, None, DEV)
    o = ops.decode_attn(q, tk, tv, cache_len + 1, Dh ** -0.5, table.col_valid_bits)
    qn = n(q)
    want = np.zeros((B, H * Dh), dtype=np.float32)
    for b in range(B):
        nk = lens[b] + 1
        keep = am[b, :nk]
        for h in range(H):
            s_ = (kcr[b, h, :nk] @ qn[b, h]) * np.float32(Dh ** -0.5)
            s_ = np.where(keep, s_, -np.inf)
            p = O.softmax(s_.astype(np.float32), -1)
            want[b, h * Dh:(h + 1) * Dh] = p @ vcr[b, h, :nk]
    check(n(o), want, dtype, "decode attention", scale_atol=2.0)


@pytest.mark.parametrize("dtype", DTYPES)
def test_generate_matches_full_forward(dtype):
    m, g = build_tiny(dtype)
    vx, lx, am, _ = batch(g, dtype)
    n_new = 5
    toks = m.generate(vx, lx, attention_mask=am, max_new_tokens=n_new, do_sample=False)
    assert toks.shape == (lx.shape[0], n_new) and toks.dtype == torch.long
    tol = 1e-5 if dtype == torch.float32 else None
    for b in range(lx.shape[0]):
        nreal = int(am[b].sum())
        ids = lx[b, :nreal]
        for step in range(n_new):
            with torch.no_grad():
                full = m(vx[b:b + 1], ids[None], attention_mask=torch.ones_like(ids)[None]).logits[0, -1].float()
            want_tok = int(full.argmax())
            got_tok = int(toks[b, step])
            if dtype == torch.float32:
                assert got_tok == want_tok, f"sample {b} step {step}: generate chose {got_tok}, full forward {want_tok}"
            else:
                # bf16: the decode path (f32 dot products, different summation order) may flip a near-tie; the chosen token
                # must be within bf16 noise of the full forward's best logit
                assert float(full[want_tok] - full[got_tok]) <= 0.05 * max(1.0, float(full.abs().max())), (b, step)
            ids = torch.cat([ids, toks[b, step:step + 1]])


@pytest.mark.parametrize("use_graph", [False, True])
@pytest.mark.parametrize("batched", [False, True])
@pytest.mark.parametrize("dtype", DTYPES)
def test_generate_reproduces_the_reference_continuation(dtype, batched, use_graph):
    """Pinned to the reference: tests/golden/tiny_generate.npz holds the greedy continuation (8 tokens per
    sample) the imported REFERENCE produced - make_golden.py::g_tiny_generate, the semantics of src/aki.py:136-209 +
    src/aki_generation.py:58-84 - with its top-1/top-2 margin per step.  `generate(do_sample=False)` must reproduce the ids:
    exactly in fp32 (smallest margin in the fixture 2.2e-3 against 1e-5 of arithmetic noise); in bf16 wherever the reference's
    margin exceeds the bf16 noise of the logits (a near-tie may flip - the comparison of that sample then stops, because the
    continuation has forked).  One sample at a time (what the reference effectively supports) and as one right-padded batch
    (build-defined: every sample continues from its own length); eager decode steps and the hipGraph replay."""
    from conftest import load_golden
    m, g = build_tiny(dtype)
    vx, lx, am, _ = batch(g, dtype)
    gg = load_golden("tiny_generate.npz")
    want, margin = gg["tokens"], gg["margin"]
    n_new = want.shape[1]
    B = lx.shape[0]
    if batched:
        got = m.generate(vx, lx, attention_mask=am, max_new_tokens=n_new, do_sample=False, eos_token_id=[], use_graph=use_graph).cpu().numpy()
    else:
        rows = []
        for b in range(B):
            nreal = int(am[b].sum())
            rows.append(m.generate(vx[b:b + 1], lx[b:b + 1, :nreal], attention_mask=am[b:b + 1, :nreal], max_new_tokens=n_new,
                                   do_sample=False, eos_token_id=[], use_graph=use_graph).cpu().numpy()[0])
        got = np.stack(rows)
    assert got.shape == want.shape
    noise = 0.0 if dtype == torch.float32 else 0.05          # bf16: the reference's own bf16 run is off by up to 0.024 per logit (tiny_e2e.npz)
    compared = 0
    for b in range(B):
        for t_ in range(n_new):
            if margin[b, t_] <= noise:
                if got[b, t_] != want[b, t_]:
                    break                                     # forked at a near-tie: later tokens are not comparable
                continue
            assert got[b, t_] == want[b, t_], f"sample {b} step {t_}: generate chose {got[b, t_]}, the reference {want[b, t_]} (margin {margin[b, t_]:.4f})"
            compared += 1
    assert compared >= (B * n_new if dtype == torch.float32 else B)


def test_generate_decode_logits_match_full_forward_fp32():
    """Tighter check on the numbers themselves (fp32): logits of decode_step == logits of the full forward."""
    m, g = build_tiny(torch.float32)
    vx, lx, am, _ = batch(g, torch.float32)
    b = 0
    nreal = int(am[b].sum())
    ids = lx[b:b + 1, :nreal]
    with torch.no_grad():
        vt = m.vision_tokenizer(m._encode_vision_x(vx[b:b + 1]))
        prep = m._prepare_inputs_for_forward(vision_tokens=vt, lang_x=ids, attention_mask=torch.ones_like(ids), padding_side="right")
        out = m.lang_model(inputs_embeds=prep["inputs_embeds"], attention_mask=prep["attention_mask"], use_cache=True, cache_capacity=64)
        cache = out.past_key_values
        nxt = out.logits[0, -1].argmax()[None]
        for step in range(4):
            dec = m.lang_model.decode_step(input_ids=nxt, past_key_values=cache)[0]
            ids = torch.cat([ids, nxt[None]], dim=1)
            full = m(vx[b:b + 1], ids, attention_mask=torch.ones_like(ids)).logits[0, -1]
            err = (dec.float() - full.float()).abs().max().item()
            from conftest import record_parity
            record_parity("fp32 decode step vs full forward over the extended prompt", torch.float32, err, (dec.float() - full.float()).abs().mean().item(),
                          full.abs().max().item(), "1e-5*max(1,max|ref|)")
            assert err <= 1e-5 * max(1.0, full.abs().max().item()), f"step {step}: decode vs full forward max err {err:.3g}"
            nxt = dec.argmax()[None]
    assert cache.get_seq_length() == prep["inputs_embeds"].shape[1] + 4


def test_generate_stops_at_eos_and_pads():
    m, g = build_tiny(torch.float32)
    vx, lx, am, _ = batch(g, torch.float32)
    first = m.generate(vx, lx, attention_mask=am, max_new_tokens=3)
    eos = int(first[0, 0])          # declare sample 0's first token to be EOS
    toks = m.generate(vx, lx, attention_mask=am, max_new_tokens=6, eos_token_id=eos, pad_token_id=gen.TINY["pad_token_id"])
    assert int(toks[0, 0]) == eos and bool((toks[0, 1:] == gen.TINY["pad_token_id"]).all())
    with pytest.raises(NotImplementedError):
        m.generate(vx, lx, attention_mask=am, num_beams=4, num_return_sequences=2)


def test_generate_default_eos_comes_from_the_language_model_config():
    """The reference's callers pass only max_new_tokens / do_sample (local_demo.py:76-87, eval_cv_bench/eval.py:99-104) and
    HF `generate` stops on generation_config.eos_token_id: so must the drop-in, without an eos_token_id argument."""
    from types import SimpleNamespace
    m, g = build_tiny(torch.float32)
    vx, lx, am, _ = batch(g, torch.float32)
    pad = gen.TINY["pad_token_id"]
    free = m.generate(vx, lx, attention_mask=am, max_new_tokens=6, eos_token_id=[])       # explicit "no EOS": all 6 tokens
    assert free.shape[1] == 6
    m.lang_model.config.eos_token_id = int(free[0, 1])         # config.json style (one id)
    assert m.default_eos_token_ids() == [int(free[0, 1])]
    toks = m.generate(vx, lx, attention_mask=am, max_new_tokens=6)
    assert toks[0, :2].tolist() == free[0, :2].tolist() and bool((toks[0, 2:] == pad).all())
    # generation_config.json style (a list) takes precedence, like HF
    m.lang_model.generation_config = SimpleNamespace(eos_token_id=sorted(set(free[:, 0].tolist())))
    toks = m.generate(vx, lx, attention_mask=am, max_new_tokens=6)
    assert toks.shape[1] == 1 and toks[:, 0].tolist() == free[:, 0].tolist()


def test_decode_past_cache_capacity_raises():
    """decoding past the cache capacity used to write into the next (batch, head) slab silently."""
    from aki_amd import AkiError
    from aki_amd.phi3 import DecodeGraph
    m, g = build_tiny(torch.float32)
    vx, lx, am, _ = batch(g, torch.float32)
    with torch.no_grad():
        prep = m._prepare_inputs_for_forward(vision_tokens=m.vision_tokenizer(m._encode_vision_x(vx)), lang_x=lx, attention_mask=am,
                                             padding_side="right")
        L = prep["inputs_embeds"].shape[1]
        out = m.lang_model(inputs_embeds=prep["inputs_embeds"], attention_mask=prep["attention_mask"], use_cache=True, cache_capacity=L + 2)
        cache = out.past_key_values
        nxt = out.logits[:, -1].argmax(-1)
        for _ in range(2):
            nxt = m.lang_model.decode_step(input_ids=nxt, past_key_values=cache).argmax(-1)
        guard = cache.k[0].clone()
        with pytest.raises(AkiError, match="KV cache is full"):
            m.lang_model.decode_step(input_ids=nxt, past_key_values=cache)
        with pytest.raises(AkiError, match="KV cache is full"):
            DecodeGraph(m.lang_model, cache).step(nxt)
        assert torch.equal(guard, cache.k[0]) and cache.host_len == L + 2


@pytest.mark.parametrize("dtype", DTYPES)
def test_forward_continues_from_past_key_values(dtype):
    """`model(vision_x=None, lang_x=new_ids, past_key_values=cache)` - the continuation call of the reference
    (src/vlm.py:463-475, src/aki.py:125-130): logits of the new tokens == a fresh full forward over prompt + new tokens."""
    m, g = build_tiny(dtype)
    vx, lx, am, _ = batch(g, dtype)
    b = 0
    nreal = int(am[b].sum())
    ids = lx[b:b + 1, :nreal]
    new = torch.tensor([[17, 93, 5]], device=DEV)
    with torch.no_grad():
        first = m(vx[b:b + 1], ids, attention_mask=torch.ones_like(ids), use_cache=True)
        cache = first.past_key_values
        past_len = cache.get_seq_length()
        mask = torch.ones((1, past_len + new.shape[1]), dtype=torch.long, device=DEV)
        cont = m(None, new, attention_mask=mask, past_key_values=cache)
        full = m(vx[b:b + 1], torch.cat([ids, new], 1), attention_mask=torch.ones((1, nreal + 3), dtype=torch.long, device=DEV))
        with pytest.raises(AssertionError):
            m(None, new, attention_mask=mask[:, :-1], past_key_values=cache)
    assert cont.logits.shape == (1, 3, full.logits.shape[-1]) and cont.past_key_values is cache
    assert cache.get_seq_length() == past_len + 3
    err = (cont.logits.float() - full.logits[:, -3:].float()).abs().max().item()
    tol = (1e-5 if dtype == torch.float32 else 3e-2) * max(1.0, full.logits.float().abs().max().item())
    assert err <= tol, f"continuation vs full forward: {err:.3g} > {tol:.3g}"


@pytest.mark.parametrize("dtype", DTYPES)
def test_generate_graph_replay_equals_eager(dtype):
    """hipGraph replay of the decode step must give exactly the tokens of eager launches (same kernels, same order)."""
    m, g = build_tiny(dtype)
    vx, lx, am, _ = batch(g, dtype)
    eager = m.generate(vx, lx, attention_mask=am, max_new_tokens=9, use_graph=False)
    graph = m.generate(vx, lx, attention_mask=am, max_new_tokens=9, use_graph=True)
    assert torch.equal(eager, graph)


def test_generate_graph_greedy_stops_at_eos_like_the_eager_loop():
    """bf16, hipGraph: the greedy pick runs inside the replayed step and the host looks at the finished flags every 8th token - the
    returned tokens (pads behind an EOS, truncation at the step where the last row finished) must equal the eager loop's, which syncs
    on every token like HF's (GenerationMixin greedy branch)."""
    m, g = build_tiny(torch.bfloat16)
    vx, lx, am, _ = batch(g, torch.bfloat16)
    pad = gen.TINY["pad_token_id"]
    free = m.generate(vx, lx, attention_mask=am, max_new_tokens=20, eos_token_id=[], use_graph=False)
    assert free.shape[1] == 20
    graph_free = m.generate(vx, lx, attention_mask=am, max_new_tokens=20, eos_token_id=[], use_graph=True)
    assert torch.equal(free, graph_free)
    for eos in ([int(free[0, 3])], [int(free[0, 11]), int(free[-1, 2])], sorted(set(free[:, 5].tolist()))):
        eager = m.generate(vx, lx, attention_mask=am, max_new_tokens=20, eos_token_id=eos, pad_token_id=pad, use_graph=False)
        graph = m.generate(vx, lx, attention_mask=am, max_new_tokens=20, eos_token_id=eos, pad_token_id=pad, use_graph=True)
        assert eager.shape == graph.shape and torch.equal(eager, graph), (eos, eager.shape, graph.shape)


def test_greedy_pick_kernel_against_torch():
    """aki_greedy_pick: argmax with torch's ordering (lower index on ties, NaN first), pad for finished rows, append at
    t = cache_len + advance - start_len, eos -> done / done_at, cache_len advance; strided logits rows."""
    from aki_amd import ops
    g = torch.Generator(device="cpu").manual_seed(3)
    B, V, ld = 5, 32064, 32064 + 64
    buf = torch.randn(B, ld, generator=g).to(torch.bfloat16)
    buf[0, 777] = buf[0, 31000] = 9.5                       # a tie: the lower index wins
    buf[1, 12345] = float("nan")                            # NaN outranks every number
    buf[2, V:] = 50.0                                       # beyond V: never looked at
    buf[3, V - 1] = 40.0                                    # the last column counts
    buf = buf.to(DEV)
    logits = buf[:, :V]
    want = logits.float().argmax(-1)
    assert int(want[0]) == 777 and int(want[1]) == 12345 and int(want[3]) == V - 1
    eos = torch.tensor([int(want[3]), 5], dtype=torch.long, device=DEV)
    done = torch.tensor([0, 0, 1, 0, 0], dtype=torch.uint8, device=DEV)
    ids = torch.full((B,), -7, dtype=torch.long, device=DEV)
    tokens = torch.full((B, 6), -1, dtype=torch.long, device=DEV)
    cache_len = torch.tensor([10, 11, 12, 13, 14], dtype=torch.int32, device=DEV)
    start = torch.tensor([8, 11, 12, 10, 20], dtype=torch.int32, device=DEV)      # t = 3, 1, 1, 4, -5 (outside: not stored)
    done_at = torch.full((B,), -1, dtype=torch.int32, device=DEV)
    ops.greedy_pick(logits, ids, pad_token_id=99, eos_ids=eos, done=done, tokens=tokens, cache_len=cache_len, start_len=start, advance=True,
                    done_at=done_at)
    exp = want.clone()
    exp[2] = 99
    assert ids.tolist() == exp.tolist()
    assert cache_len.tolist() == [11, 12, 13, 14, 15]
    t = [3, 1, 1, 4]
    for b in range(4):
        row = tokens[b].tolist()
        assert row[t[b]] == int(exp[b]) and all(v == -1 for i, v in enumerate(row) if i != t[b])
    assert tokens[4].tolist() == [-1] * 6
    assert done.tolist() == [0, 0, 1, 1, 0] and done_at.tolist() == [-1, -1, -1, 4, -1]
    # the minimal form: ids only, unaligned width (scalar tail path)
    small = torch.randn(2, 1001, generator=g).to(torch.bfloat16).to(DEV)
    out = ops.greedy_pick(small, torch.empty(2, dtype=torch.long, device=DEV))
    assert out.tolist() == small.float().argmax(-1).tolist()


@pytest.mark.parametrize("M", [1, 8])
@pytest.mark.parametrize("act", ["none", "swiglu"])
def test_decode_linear_fused_rmsnorm(M, act):
    """aki_decode_linear_fwd == rmsnorm kernel followed by the linear kernel (and the oracle's arithmetic)."""
    from aki_amd import ops
    rng = gen.rng_for(f"declin{M}{act}")
    K, N = 3072, 1024
    x = rng.standard_normal((M, K), dtype=np.float32) * 3.0
    g = 1.0 + 0.1 * rng.standard_normal((K,), dtype=np.float32)
    w = rng.standard_normal((N, K), dtype=np.float32) * 0.05
    r = rng.standard_normal((M, N if act == "none" else N // 2), dtype=np.float32)
    dt = torch.bfloat16
    a = ops.ACT_SWIGLU if act == "swiglu" else ops.ACT_NONE
    fused = ops.decode_linear(t(x, dt), t(w, dt), t(g, dt), 1e-5, act=a, residual=t(r, dt))
    plain = ops.linear(ops.rmsnorm(t(x, dt), t(g, dt), 1e-5), t(w, dt), act=a, residual=t(r, dt))
    check(n(fused), n(plain).astype(np.float32), dt, "fused norm+gemv vs norm kernel + gemv", scale_atol=2.0)


@pytest.mark.parametrize("M,N,act", [(2, 32064, "none"), (5, 20000, "none"), (8, 9216, "none"), (3, 16384, "swiglu"), (8, 16384, "swiglu"), (8, 40000, "swiglu")])
def test_batched_decode_linear_with_the_norm_in_its_prologue(M, N, act):
    """2-8 rows: aki_decode_linear_fwd runs the skinny MFMA GEMM with the RMSNorm in its prologue (rows normalised into LDS, HF's rounding
    points) == the norm kernel followed by the skinny GEMM - over its K splits (8 / 4 / 2 waves per 16-feature tile) and the SwiGLU pairing."""
    from aki_amd import ops
    rng = gen.rng_for(f"skinnynorm{M}{N}{act}")
    K = 3072
    x = rng.standard_normal((M, K), dtype=np.float32) * 3.0
    x[M - 1] *= 0.01                                        # rows of very different scale: a shared / swapped rstd would show
    g = 1.0 + 0.1 * rng.standard_normal((K,), dtype=np.float32)
    w = rng.standard_normal((N, K), dtype=np.float32) * 0.05
    dt = torch.bfloat16
    a = ops.ACT_SWIGLU if act == "swiglu" else ops.ACT_NONE
    xt, wt, gt = t(x, dt), t(w, dt), t(g, dt)
    fused = ops.decode_linear(xt, wt, gt, 1e-5, act=a)
    xn = ops.rmsnorm(xt, gt, 1e-5)
    plain = ops.linear(xn, wt, act=a)
    # the normalised rows are the same bf16 values up to the summation order of the variance (one ulp of rstd): compare at the fused test's bar
    check(n(fused), n(plain).astype(np.float32), dt, "skinny GEMM with the norm in its prologue vs norm kernel + skinny GEMM", scale_atol=2.0)
    assert torch.equal(fused, ops.decode_linear(xt, wt, gt, 1e-5, act=a))        # and reproducible


@pytest.mark.parametrize("M", [2, 8, 16])
@pytest.mark.parametrize("shape", ["qkv", "gate_up", "down", "head"])
def test_skinny_gemm_on_e4m3_weights(M, shape):
    """2-16 rows on e4m3 weights (batched decode in the fp8 configuration): y = (x . w8) * row scale [+ residual], SwiGLU pairing, the RMSNorm in
    the prologue (2-8 rows; wide outputs and more rows through the norm launch) - against float64 arithmetic on the same e4m3 values and
    against the one-row W8 GEMV row by row."""
    from aki_amd import ops
    rng = gen.rng_for(f"skinnyw8{M}{shape}")
    N, K, act, norm, res = {"qkv": (9216, 3072, "none", True, False), "gate_up": (16384, 3072, "swiglu", True, False),
                            "down": (3072, 8192, "none", False, True), "head": (32064, 3072, "none", True, False)}[shape]
    dt = torch.bfloat16
    x = t(rng.standard_normal((M, K), dtype=np.float32) * 2.0, dt)
    x[M - 1] *= 0.05
    g = t(1.0 + 0.1 * rng.standard_normal((K,), dtype=np.float32), dt)
    w = t(rng.standard_normal((N, K), dtype=np.float32) * 0.05, dt)
    wq, ws = ops.quant_rows_fp8(w)
    n_out = N // 2 if act == "swiglu" else N
    r = t(rng.standard_normal((M, n_out), dtype=np.float32), dt) if res else None
    a = ops.ACT_SWIGLU if act == "swiglu" else ops.ACT_NONE
    kw = dict(act=a, residual=r)
    if norm:
        kw.update(rms_weight=g, eps=1e-5)
    got = ops.linear_w8(x, wq, ws, **kw)
    assert got.shape == (M, n_out) and torch.equal(got, ops.linear_w8(x, wq, ws, **kw))
    xin = ops.rmsnorm(x, g, 1e-5) if norm else x
    wd = wq.cpu().view(torch.float8_e4m3fn).to(torch.float32).to(DEV).double()           # the e4m3 values, widened on the host
    y = (xin.double() @ wd.T) * ws.double()[None, :]
    if act == "swiglu":
        y = y[:, n_out:] * torch.nn.functional.silu(y[:, :n_out])
    if res:
        y = y + r.double()
    check(n(got), y.cpu().numpy().astype(np.float32), dt, f"skinny W8 GEMM {shape} M={M}", scale_atol=2.0)
    rows = torch.cat([ops.linear_w8(x[m:m + 1], wq, ws, act=a, residual=None if r is None else r[m:m + 1],
                                    **({"rms_weight": g, "eps": 1e-5} if norm else {})) for m in range(M)])
    check(n(got), n(rows).astype(np.float32), dt, "skinny W8 GEMM vs the one-row W8 GEMV", scale_atol=2.0)


@pytest.mark.parametrize("B,H,cap,lens", [(3, 4, 300, [17, 200, 298]), (2, 2, 5000, [4100, 63]), (8, 32, 700, [655] * 8), (1, 32, 64, [0])])
def test_decode_attn_fused_vs_two_kernels(B, H, cap, lens):
    """RoPE + append + split-KV attention in one launch == rope_append followed by decode_attn; caches end up identical."""
    from aki_amd import ops
    Dh, dt = 96, torch.bfloat16
    rng = gen.rng_for(f"decfused{B}{H}{cap}")
    kc = t(rng.standard_normal((B, H, cap, Dh), dtype=np.float32), dt)
    vc = t(rng.standard_normal((B, H, cap, Dh), dtype=np.float32), dt)
    for b_, n_ in enumerate(lens):          # rows at and beyond the append position are unwritten cache: poison them (the
        kc[b_, :, n_:] = float("nan")       # kernels may touch them only with probability 0 and must not let 0 * NaN through)
        vc[b_, :, n_:] = float("nan")
    qkv = t(rng.standard_normal((B, 3 * H * Dh), dtype=np.float32), dt)
    cos, sin = O.rope_cos_sin(np.arange(cap)[None], Dh)
    tc, ts = torch.from_numpy(cos[0]).to(DEV), torch.from_numpy(sin[0]).to(DEV)
    cl = torch.tensor(lens, dtype=torch.int32, device=DEV)
    am = np.ones((B, cap), dtype=bool)
    am[0, 3:7] = False
    bits = ops.MaskTable.from_host([[(0, 0, 0, 0)]] * B, am, None, DEV).col_valid_bits
    k1, v1, k2, v2 = kc.clone(), vc.clone(), kc.clone(), vc.clone()
    q = ops.rope_append(qkv, tc, ts, cl, cl, k1, v1, H)
    want = ops.decode_attn(q, k1, v1, cl + 1, Dh ** -0.5, bits)
    ws = ops.decode_attn_workspace(B, H, Dh, cap, DEV)
    for rep in range(2):             # twice through the same workspace: the arrival counters must re-arm themselves
        k2.copy_(kc); v2.copy_(vc)
        got = ops.decode_attn_fused(qkv, tc, ts, cl, k2, v2, H, Dh ** -0.5, bits, max(lens) + 1, ws)
        assert bool(torch.isfinite(got.float()).all()) and bool(torch.isfinite(want.float()).all())
        fin = torch.isfinite(k1.float())
        assert torch.equal(fin, torch.isfinite(k2.float())) and torch.equal(k1[fin], k2[fin]) and torch.equal(v1[fin], v2[fin]), "appended rows differ"
        check(n(got), n(want).astype(np.float32), dt, f"fused decode attention (pass {rep})", scale_atol=2.0)
    assert int(ws[: B * H].abs().sum()) == 0


# ---- sampling and beam search (HF keyword arguments the reference forwards through **kwargs, src/aki.py:160-207) --------------
def _seq_logprob(m, vx_b, prompt_ids, new_tokens):
    """Sum of log-probabilities of `new_tokens` after `prompt_ids`, from FULL forwards (fp32) - the yardstick the decode path
    (KV cache, cache re-ordering) is checked against."""
    ids = prompt_ids
    total = 0.0
    for tok in new_tokens:
        with torch.no_grad():
            lg = m(vx_b, ids[None], attention_mask=torch.ones_like(ids)[None]).logits[0, -1].float()
        total += float(torch.log_softmax(lg, -1)[tok])
        ids = torch.cat([ids, torch.tensor([tok], device=ids.device)])
    return total


def test_sampling_modes():
    from aki_amd.aki import sample_next
    m, g = build_tiny(torch.float32)
    vx, lx, am, _ = batch(g, torch.float32)
    greedy = m.generate(vx, lx, attention_mask=am, max_new_tokens=4, eos_token_id=[])
    # top_k = 1 is greedy, whatever the temperature
    tk1 = m.generate(vx, lx, attention_mask=am, max_new_tokens=4, eos_token_id=[], do_sample=True, top_k=1, temperature=0.7)
    assert torch.equal(tk1, greedy)
    # a vanishing nucleus keeps only the most probable token
    tp = m.generate(vx, lx, attention_mask=am, max_new_tokens=4, eos_token_id=[], do_sample=True, top_p=1e-6)
    assert torch.equal(tp, greedy)
    # same generator state -> same draw; the draw respects top_k
    g1 = torch.Generator(device=DEV).manual_seed(5)
    g2 = torch.Generator(device=DEV).manual_seed(5)
    a = m.generate(vx, lx, attention_mask=am, max_new_tokens=6, eos_token_id=[], do_sample=True, top_k=5, generator=g1)
    b = m.generate(vx, lx, attention_mask=am, max_new_tokens=6, eos_token_id=[], do_sample=True, top_k=5, generator=g2)
    assert torch.equal(a, b) and a.shape == (lx.shape[0], 6)
    # the step itself: frequencies follow the filtered softmax
    logits = torch.tensor([[2.0, 1.0, 0.0, -1.0, -5.0]], device=DEV).repeat(20000, 1)
    draws = sample_next(logits, temperature=1.0, top_k=3, generator=torch.Generator(device=DEV).manual_seed(0))
    freq = torch.bincount(draws, minlength=5).float() / draws.numel()
    want = torch.softmax(torch.tensor([2.0, 1.0, 0.0]), -1)
    assert float(freq[3:].sum()) == 0.0 and torch.allclose(freq[:3].cpu(), want, atol=0.015)
    draws = sample_next(logits, top_p=0.7, generator=torch.Generator(device=DEV).manual_seed(1))       # mass 0.63 + 0.23: two tokens
    assert set(draws.unique().tolist()) == {0, 1}
    with pytest.raises(NotImplementedError):
        m.generate(vx, lx, attention_mask=am, max_new_tokens=2, num_beams=2, do_sample=True)


@pytest.mark.parametrize("K", [2, 3])
def test_beam_search_against_full_forward_scores(K):
    """The returned hypothesis must (a) score, by full forwards, what beam search says; (b) score at least as well as the greedy
    continuation of the same length; (c) equal an independent beam search that uses full forwards only (no KV cache, no
    re-ordering) - which pins AkiKVCache.select_rows and the expansion of the prompt rows to beams."""
    m, g = build_tiny(torch.float32)
    vx, lx, am, _ = batch(g, torch.float32)
    n_new = 4
    got = m.generate(vx, lx, attention_mask=am, max_new_tokens=n_new, num_beams=K, eos_token_id=[])
    greedy = m.generate(vx, lx, attention_mask=am, max_new_tokens=n_new, eos_token_id=[])
    assert got.shape == (lx.shape[0], n_new)
    for b in range(lx.shape[0]):
        prompt = lx[b, : int(am[b].sum())]
        # reference beam search on full forwards
        beams = [(0.0, [])]
        for t in range(n_new):
            cand = []
            for sc, toks in beams:
                ids = torch.cat([prompt, torch.tensor(toks, dtype=torch.long, device=DEV)])
                with torch.no_grad():
                    lg = m(vx[b:b + 1], ids[None], attention_mask=torch.ones_like(ids)[None]).logits[0, -1].float()
                lp = torch.log_softmax(lg, -1)
                top = lp.topk(2 * K)
                cand += [(sc + float(v), toks + [int(i)]) for v, i in zip(top.values, top.indices)]
            cand.sort(key=lambda c: -c[0])
            beams = cand[:K]
        want = beams[0][1]
        assert got[b].tolist() == want, f"sample {b}: beam search {got[b].tolist()} vs full-forward beam search {want}"
        s_beam = _seq_logprob(m, vx[b:b + 1], prompt, got[b].tolist())
        s_greedy = _seq_logprob(m, vx[b:b + 1], prompt, greedy[b].tolist())
        assert abs(s_beam - beams[0][0]) < 1e-3 and s_beam >= s_greedy - 1e-4


def test_beam_search_stops_on_eos_and_pads():
    m, g = build_tiny(torch.float32)
    vx, lx, am, _ = batch(g, torch.float32)
    free = m.generate(vx, lx, attention_mask=am, max_new_tokens=3, num_beams=2, eos_token_id=[])
    eos = sorted(set(free[:, 1].tolist()))                    # the second token of every sample's best beam ends it
    out = m.generate(vx, lx, attention_mask=am, max_new_tokens=6, num_beams=2, eos_token_id=eos, pad_token_id=0)
    assert out.shape[1] <= 6
    for b in range(lx.shape[0]):
        row = out[b].tolist()
        hits = [i for i, t_ in enumerate(row) if t_ in eos]
        if hits:                                              # everything behind the first EOS is padding
            assert all(t_ == 0 for t_ in row[hits[0] + 1:])


# ---- the one-launch decode step (decode_chain.hip) -----------------------------------------------------------------------
def _full_width_lm(n_layers, seed=0):
    from aki_amd.phi3 import Phi3ForCausalLM, make_phi3_config
    torch.manual_seed(seed)
    cfg = make_phi3_config(num_hidden_layers=n_layers, vocab_size=4096, pad_token_id=0, eos_token_id=2)
    lm = Phi3ForCausalLM(cfg)
    g = torch.Generator().manual_seed(seed)
    for n, p in lm.named_parameters():
        if p.dim() == 1:
            p.data.copy_(1.0 + 0.2 * torch.randn(p.shape, generator=g))          # non-unit RMSNorm gains
        else:
            p.data.copy_(torch.randn(p.shape, generator=g) * 0.02)
    return lm.to(DEV).to(torch.bfloat16).eval(), cfg


@pytest.mark.parametrize("prompt,steps,fp8", [(655, 6, False), (70, 70, False), (4200, 3, False), (655, 4, True)])
def test_decode_chain_is_bit_identical_to_the_per_layer_launches(prompt, steps, fp8):
    """decode_chain.hip runs every layer of a batch-1 decode step in ONE launch; its arithmetic is decode.hip's row for row, so the
    logits of every step - and the K/V rows it appends - equal the five-launch-per-layer path's bit for bit, at Phi-3.5-mini's
    width (3 layers), across a 64-key tile boundary of the cache (70 + 70 steps), at a long context where the attention splits
    over 66+ workgroups per head, and with e4m3 weights (the fp8 configuration's weight-only GEMVs).  The chain's sticky error
    word (a dependency wait that gave up) must stay 0."""
    from aki_amd import ops
    lm, cfg = _full_width_lm(3)
    if fp8:
        lm.enable_fp8()
    x = (torch.randn(1, prompt, cfg.hidden_size, generator=torch.Generator().manual_seed(1)) * 0.5).to(torch.bfloat16).to(DEV)
    am = np.ones((1, prompt), dtype=bool)
    am[0, 3:9] = False                                        # a hole in the prompt: the valid-column bits are honoured
    table = ops.MaskTable.from_host([[(4, 40, 40, prompt - 8)]], am, [prompt], DEV)
    outs = {}
    for chained in (False, True):
        lm.model.use_decode_chain = chained
        with torch.no_grad():
            out = lm(inputs_embeds=x, attention_mask=table, use_cache=True, cache_capacity=prompt + steps + 3)
            cache = out.past_key_values
            ids = out.logits[:, -1].float().argmax(-1)
            logits = []
            for _ in range(steps):
                lg = lm.decode_step(input_ids=ids, past_key_values=cache)
                logits.append(lg.clone())
                ids = lg.float().argmax(-1)
        chain = getattr(cache, "chain", None)
        assert (chain is not None) == chained
        if chained:
            assert chain.error_code() == 0
        outs[chained] = (torch.stack(logits), [k[:, :, : prompt + steps].clone() for k in cache.k], [v[:, :, : prompt + steps].clone() for v in cache.v])
    lm.model.use_decode_chain = True
    a, b = outs[False], outs[True]
    assert bool(torch.isfinite(a[0].float()).all())
    assert torch.equal(a[0], b[0]), f"{int((a[0] != b[0]).sum())} logits differ over {steps} steps"
    for ka, kb in zip(a[1] + a[2], b[1] + b[2]):
        assert torch.equal(ka, kb)


@pytest.mark.parametrize("B,prompt,steps", [(2, 200, 5), (8, 655, 4), (5, 70, 70), (3, 4200, 3)])
def test_batched_decode_chain_is_bit_identical_to_the_per_layer_launches(B, prompt, steps):
    """The batched chain (2..8 sequences per step: 16-feature MFMA tiles instead of dot-product rows, attention items per sequence) against
    the five-launch-per-layer batched path: logits of every step and the appended K/V rows bit for bit - ragged prompt lengths, a hole in
    one prompt, across a 64-key tile boundary (70 + 70 steps), at a long context - and the error word stays 0."""
    from aki_amd import ops
    lm, cfg = _full_width_lm(3)
    x = (torch.randn(B, prompt, cfg.hidden_size, generator=torch.Generator().manual_seed(1)) * 0.5).to(torch.bfloat16).to(DEV)
    lens = [prompt - 7 * b for b in range(B)]
    am = np.zeros((B, prompt), dtype=bool)
    for b, n_ in enumerate(lens):
        am[b, :n_] = True
    am[B - 1, 3:9] = False
    table = ops.MaskTable.from_host([[(4, 40, 40, n_ - 8)] for n_ in lens], am, lens, DEV)
    outs = {}
    import contextlib
    from aki_amd import _lib
    for chained in (False, True):
        lm.model.use_decode_chain_batched = chained
        # the batched chain lives in the lab library only (round 6: built, bit-identical, not faster - out of the product)
        with (_lib.use_lab(0) if chained else contextlib.nullcontext()), torch.no_grad():
            out = lm(inputs_embeds=x, attention_mask=table, use_cache=True, cache_capacity=prompt + steps + 3)
            cache = out.past_key_values
            ids = out.logits[:, -1].float().argmax(-1)
            logits = []
            for _ in range(steps):
                lg = lm.decode_step(input_ids=ids, past_key_values=cache)
                logits.append(lg.clone())
                ids = lg.float().argmax(-1)
            chain = getattr(cache, "chain", None)
            assert (chain is not None) == chained
            if chained:
                assert chain.batch == B and chain.error_code() == 0
        # the rows each sequence holds: its prompt + the appended steps (what lies beyond is unwritten cache)
        rows = lambda c: [torch.cat([c_[b_, :, : lens[b_] + steps].reshape(-1) for b_ in range(B)]) for c_ in c]
        outs[chained] = (torch.stack(logits), rows(cache.k), rows(cache.v), cache.cache_len.clone())
    lm.model.use_decode_chain_batched = False
    a, b = outs[False], outs[True]
    assert bool(torch.isfinite(a[0].float()).all()) and torch.equal(a[3], b[3])
    assert torch.equal(a[0], b[0]), f"{int((a[0] != b[0]).sum())} logits differ over {steps} steps"
    for ka, kb in zip(a[1] + a[2], b[1] + b[2]):
        assert torch.equal(ka.view(torch.int16), kb.view(torch.int16))


def test_decode_chain_graph_replay_and_poisoned_workspace():
    """The chained step under hipGraph replay (its counter memset is a graph node) equals eager steps bit for bit, and a workspace
    whose hand-off vectors are poisoned between steps changes nothing: every byte a phase reads was written in the same step."""
    from aki_amd import ops
    from aki_amd.phi3 import DecodeGraph
    lm, cfg = _full_width_lm(2, seed=3)
    prompt, steps = 200, 5
    x = (torch.randn(1, prompt, cfg.hidden_size, generator=torch.Generator().manual_seed(2)) * 0.5).to(torch.bfloat16).to(DEV)
    table = ops.MaskTable.from_host([[(4, 100, 100, 180)]], np.ones((1, prompt), dtype=bool), [prompt], DEV)
    res = {}
    for mode in ("eager", "graph"):
        with torch.no_grad():
            out = lm(inputs_embeds=x, attention_mask=table, use_cache=True, cache_capacity=prompt + steps + 8)
            cache = out.past_key_values
            ids = out.logits[:, -1].float().argmax(-1)
            stepper = DecodeGraph(lm, cache) if mode == "graph" else None
            logits = []
            for i in range(steps):
                lg = stepper.step(ids) if stepper else lm.decode_step(input_ids=ids, past_key_values=cache)
                logits.append(lg.clone())
                ids = lg.float().argmax(-1)
                if mode == "eager":                              # poison everything behind the counters and the error word
                    torch.cuda.synchronize()
                    ch = cache.chain
                    lo = ch.err_index * 4 + 256
                    ch.ws[lo:].fill_(0xFF)
        assert cache.chain.error_code() == 0
        res[mode] = torch.stack(logits)
    assert torch.equal(res["eager"], res["graph"])


@pytest.mark.parametrize("chained,batch,prompt,cap", [(True, 1, 200, 4400), (False, 1, 200, 4400), (False, 8, 150, 700)])
def test_decode_replayed_steps_equal_eager_steps_in_a_cache_larger_than_4096_keys(chained, batch, prompt, cap):
    """A captured decode step sizes its attention launch for the whole cache, an eager one for the keys cached so far.  Beyond 2048
    (head, tile) items the split takes two tiles per item: that count now comes from the CAPACITY in both cases, so the two launches
    cut the keys at the same places and differ only by trailing empty items - eager and replayed logits are equal bit for bit.  (A
    4096-token soak of the chain, tools/decode_chain_soak.py, found the replayed continuation leaving the eager one at token 128 when the
    count still came from the keys cached so far.)"""
    from aki_amd import ops
    from aki_amd.phi3 import DecodeGraph
    lm, cfg = _full_width_lm(2, seed=9)
    lm.model.use_decode_chain = chained
    steps = 5                      # (batch 8: 8 x 32 heads x 11 tiles of a 700-key cache are already more than 2048 items)
    x = (torch.randn(batch, prompt, cfg.hidden_size, generator=torch.Generator().manual_seed(2)) * 0.5).to(torch.bfloat16).to(DEV)
    table = ops.MaskTable.from_host([[(4, 40, 40, prompt - 8)]] * batch, np.ones((batch, prompt), dtype=bool), [prompt] * batch, DEV)
    res = {}
    with torch.no_grad():
        for mode in ("eager", "graph"):
            out = lm(inputs_embeds=x, attention_mask=table, use_cache=True, cache_capacity=cap)
            cache, ids = out.past_key_values, out.logits[:, -1].float().argmax(-1)
            st = DecodeGraph(lm, cache) if mode == "graph" else None
            logits = []
            for _ in range(steps):
                lg = st.step(ids) if st is not None else lm.decode_step(input_ids=ids, past_key_values=cache)
                logits.append(lg.clone())
                ids = lg.float().argmax(-1)
            res[mode] = torch.stack(logits)
            assert (getattr(cache, "chain", None) is not None) == chained
    lm.model.use_decode_chain = True
    assert torch.equal(res["eager"], res["graph"]), f"{int((res['eager'] != res['graph']).sum())} logits differ between eager and replayed steps"


def test_generate_on_the_decode_chain_equals_generate_on_the_captured_five_launch_steps():
    """`AKI.generate`, one sample, a full-width 2-layer decoder behind a small vision tower: with the one-launch decode chain the greedy
    loop runs eagerly (five launches per token, no graph), without it the steps are captured into a hipGraph after the first eager one -
    two code paths in aki.py::generate.  Logits are bit-identical between the two decode implementations, so the tokens must be equal:
    all of them, and with an EOS in the middle (pads behind it, truncation)."""
    from aki_amd.factory import build_aki
    from aki_amd.phi3 import make_phi3_config
    from aki_amd.siglip import make_siglip_config
    m = build_aki(lm_config=make_phi3_config(num_hidden_layers=2), vis_config=make_siglip_config(num_hidden_layers=1, image_size=224),
                  dtype=torch.bfloat16, device=DEV, seed=3).eval()
    g = torch.Generator(device="cpu").manual_seed(5)
    n_txt = 40
    ids = torch.randint(3, 32000, (1, n_txt), generator=g)
    ids[0, 0], ids[0, 6] = 1, m.media_token_id
    vx = ((torch.rand((1, 1, 1, 3, 224, 224), generator=g) - 0.5) / 0.5).to(DEV, torch.bfloat16)
    ids, am = ids.to(DEV), torch.ones(1, n_txt, dtype=torch.long, device=DEV)
    outs = {}
    for chained in (True, False):
        m.lang_model.model.use_decode_chain = chained
        outs[chained] = m.generate(vx, ids, attention_mask=am, max_new_tokens=24, do_sample=False, eos_token_id=[])
    assert outs[True].shape == (1, 24) and torch.equal(outs[True], outs[False])
    eos = [int(outs[True][0, 13])]
    first = int((outs[True][0] == eos[0]).nonzero()[0])
    for chained in (True, False):
        m.lang_model.model.use_decode_chain = chained
        got = m.generate(vx, ids, attention_mask=am, max_new_tokens=24, do_sample=False, eos_token_id=eos)
        assert got.shape == (1, first + 1) and torch.equal(got[0], outs[True][0, : first + 1]), (chained, got.shape, first)
    m.lang_model.model.use_decode_chain = True


@pytest.mark.parametrize("fp8", [False, True])
def test_decode_chain_full_depth_graph_replay_equals_the_five_launch_path(fp8):
    """All 32 layers at Phi-3.5-mini's width, hipGraph replay, 10 greedy steps, bf16 and e4m3 weights (each has its own instance of the
    chain kernel): the logits of every replayed step equal the five-launch-per-layer path bit for bit.  (This is the check round 4 was missing when the chain's arrival counters were zeroed
    by a captured hipMemsetAsync node: under replay the block held a constant non-zero pattern, no workgroup waited for its
    producers, and the step ran at the bare weight stream's speed with wrong logits - while a 2-layer graph test passed.)"""
    from aki_amd import ops
    from aki_amd.phi3 import DecodeGraph
    lm, cfg = _full_width_lm(32, seed=5)
    if fp8:
        lm.enable_fp8()
    prompt, steps = 655, 10
    x = (torch.randn(1, prompt, cfg.hidden_size, generator=torch.Generator().manual_seed(4)) * 0.5).to(torch.bfloat16).to(DEV)
    table = ops.MaskTable.from_host([[(6, 150, 150, 638)]], np.ones((1, prompt), dtype=bool), [prompt], DEV)
    res = {}
    for chained in (False, True):
        lm.model.use_decode_chain = chained
        with torch.no_grad():
            out = lm(inputs_embeds=x, attention_mask=table, use_cache=True, cache_capacity=prompt + steps + 8)
            cache = out.past_key_values
            ids = out.logits[:, -1].float().argmax(-1)
            stepper = DecodeGraph(lm, cache)
            logits = []
            for _ in range(steps):
                lg = stepper.step(ids)
                logits.append(lg.clone())
                ids = lg.float().argmax(-1)
        if chained:
            assert cache.chain is not None and cache.chain.error_code() == 0
        res[chained] = torch.stack(logits)
    lm.model.use_decode_chain = True
    bad = [int((res[True][i] != res[False][i]).sum()) for i in range(steps)]
    assert sum(bad) == 0, f"logits differing from the five-launch path per replayed step: {bad}"


def test_greedy_pick_gathers_the_next_steps_embedding_row():
    """aki_greedy_pick_embed: the picked token's row of DecoupledEmbedding's two tables (ids above max_original_id index the additional one,
    src/helpers.py:440-492) lands in next_embeds in the same launch - bit for bit the module's own forward; finished rows gather the pad row."""
    from aki_amd import ops
    from aki_amd.helpers import DecoupledEmbedding
    g = torch.Generator(device="cpu").manual_seed(11)
    n_orig, n_add, d, B = 500, 3, 3072, 6
    emb = DecoupledEmbedding(max_original_id=n_orig - 1, num_additional_embeddings=n_add, num_original_embeddings=n_orig, embedding_dim=d,
                             pad_token_id=0).to(DEV, torch.bfloat16)
    with torch.no_grad():
        emb.weight.copy_(torch.randn(n_orig, d, generator=g))
        emb.additional_embedding.weight.copy_(torch.randn(n_add, d, generator=g))
    V = n_orig + n_add
    logits = torch.randn(B, V, generator=g).to(torch.bfloat16)
    logits[0, n_orig + 2] = 30.0                            # additional table, last row
    logits[1, n_orig] = 30.0                                # additional table, first row
    logits[2, n_orig - 1] = 30.0                            # last original row
    logits[3, 17] = 30.0
    logits = logits.to(DEV)
    done = torch.tensor([0, 0, 0, 0, 1, 0], dtype=torch.uint8, device=DEV)       # row 4 is finished: pad (= 7 here)
    ids = torch.empty(B, dtype=torch.long, device=DEV)
    out = torch.full((B, d), 7.0, dtype=torch.bfloat16, device=DEV)
    ops.greedy_pick(logits, ids, pad_token_id=7, done=done, embed=(emb.weight, emb.additional_embedding.weight, emb.max_original_id), next_embeds=out)
    want_ids = logits.float().argmax(-1)
    want_ids[4] = 7
    assert ids.tolist() == want_ids.tolist() and ids[:4].tolist() == [n_orig + 2, n_orig, n_orig - 1, 17]
    assert torch.equal(out, emb(ids))
    # one table (plain nn.Embedding)
    ops.greedy_pick(logits[:, :n_orig].contiguous(), ids, embed=(emb.weight, None, n_orig - 1), next_embeds=out)
    assert torch.equal(out, torch.nn.functional.embedding(ids, emb.weight))
    with pytest.raises(ops.AkiError):                       # more logit columns than embedding rows
        ops.greedy_pick(logits, ids, embed=(emb.weight, None, n_orig - 1), next_embeds=out)


def _tiny_full_width_aki():
    from aki_amd.factory import build_aki
    from aki_amd.phi3 import make_phi3_config
    from aki_amd.siglip import make_siglip_config
    m = build_aki(lm_config=make_phi3_config(num_hidden_layers=2), vis_config=make_siglip_config(num_hidden_layers=1, image_size=224),
                  dtype=torch.bfloat16, device=DEV, seed=3).eval()
    g = torch.Generator(device="cpu").manual_seed(5)
    n_txt = 40
    ids = torch.randint(3, 32000, (1, n_txt), generator=g)
    ids[0, 0], ids[0, 6] = 1, m.media_token_id
    vx = ((torch.rand((1, 1, 1, 3, 224, 224), generator=g) - 0.5) / 0.5).to(DEV, torch.bfloat16)
    return m, vx, ids.to(DEV), torch.ones(1, n_txt, dtype=torch.long, device=DEV)


@pytest.mark.parametrize("skip", [0, 5, 11, 17])
@pytest.mark.parametrize("loop", ["greedy", "greedy_eos", "plain"])
def test_generate_recovers_from_a_decode_chain_give_up(skip, loop):
    """VERDICT r4 / ADVICE r4: a dependency wait of the one-launch decode chain that gives up leaves garbage in that step and a sticky error
    word - `generate` must never return such tokens.  The lab library makes one wait of chain launch number `skip` give up at once (error
    word set, every flag raised, the rest of the launch runs on stale inputs): `generate` has to notice at its next synchronisation point,
    warn, decode the unverified tokens again on the five-launch-per-layer path and return exactly what a chain-free run returns - in the
    greedy loop (with and without an EOS check: the every-8th-token check vs only the final one) and in the plain (sampling-capable) loop."""
    from aki_amd import _lib
    m, vx, ids, am = _tiny_full_width_aki()
    kw = dict(max_new_tokens=24, do_sample=False)
    if loop == "plain":
        kw["use_graph"] = False
    m.lang_model.model.use_decode_chain = False
    ref_free = m.generate(vx, ids, attention_mask=am, eos_token_id=[], **kw)
    kw["eos_token_id"] = [int(ref_free[0, 19])] if loop != "greedy" else []
    want = m.generate(vx, ids, attention_mask=am, **kw)
    m.lang_model.model.use_decode_chain = True
    with _lib.use_lab(0) as lab:
        clean = m.generate(vx, ids, attention_mask=am, **kw)                    # the chain, no fault: the same tokens, no warning
        assert torch.equal(clean, want)
        lab.aki_lab_set_chain_fault((1 << 8) | 3, skip)                         # layer 1's o_proj wait of launch `skip`
        with pytest.warns(RuntimeWarning, match="decode chain"):
            got = m.generate(vx, ids, attention_mask=am, **kw)
        lab.aki_lab_set_chain_fault(0, 0)
    assert got.shape == want.shape and torch.equal(got, want), (got.tolist(), want.tolist())


@pytest.mark.parametrize("skip", [3, 11, 17])
def test_batched_generate_on_the_batched_chain_recovers_from_a_give_up(skip):
    """The opt-in batched chain under `generate` (two samples, right-padded, different lengths): the same tokens as the chain-free run, and
    after an injected give-up the unverified tokens are decoded again on the per-layer path - with the finished flags REBUILT from the
    verified tokens: one row has hit its EOS before the fault, the other has not (the one-sequence recovery could simply clear them)."""
    from aki_amd import _lib
    m, vx, ids, am = _tiny_full_width_aki()
    g = torch.Generator(device="cpu").manual_seed(9)
    ids2 = torch.randint(3, 32000, (1, ids.shape[1]), generator=g).to(DEV)
    ids2[0, 0], ids2[0, 6] = 1, m.media_token_id
    am2 = am.clone()
    am2[0, -5:] = 0                                                              # the second sample is five tokens shorter
    vx2 = ((torch.rand((1, 1, 1, 3, 224, 224), generator=g) - 0.5) / 0.5).to(DEV, torch.bfloat16)
    VX, IDS, AM = torch.cat([vx, vx2]), torch.cat([ids, ids2]), torch.cat([am, am2])
    kw = dict(max_new_tokens=24, do_sample=False)
    lmm = m.lang_model.model
    lmm.use_decode_chain = False
    free = m.generate(VX, IDS, attention_mask=AM, eos_token_id=[], **kw)
    row1 = set(free[1].tolist())
    eos = next(int(t_) for t_ in free[0, 2:10].tolist() if int(t_) not in row1)   # row 0 finishes within its first ten tokens, row 1 never does
    kw["eos_token_id"] = [eos]
    want = m.generate(VX, IDS, attention_mask=AM, **kw)
    lmm.use_decode_chain, lmm.use_decode_chain_batched = True, True
    try:
        with _lib.use_lab(0) as lab:
            clean = m.generate(VX, IDS, attention_mask=AM, **kw)
            assert torch.equal(clean, want)
            lab.aki_lab_set_chain_fault((1 << 8) | 3, skip)
            with pytest.warns(RuntimeWarning, match="decode chain"):
                got = m.generate(VX, IDS, attention_mask=AM, **kw)
            lab.aki_lab_set_chain_fault(0, 0)
    finally:
        lmm.use_decode_chain_batched = False
    assert got.shape == want.shape and torch.equal(got, want), (got.tolist(), want.tolist())


def test_forward_from_past_key_values_recovers_from_a_decode_chain_give_up():
    """The reference's `past_key_values is not None` call (src/vlm.py:463-475) runs T teacher-forced decode steps: same guarantee."""
    from aki_amd import _lib, ops
    lm, cfg = _full_width_lm(2, seed=7)
    g = torch.Generator().manual_seed(2)
    x = (torch.randn(1, 70, cfg.hidden_size, generator=g) * 0.5).to(torch.bfloat16).to(DEV)
    new = torch.randint(3, cfg.vocab_size, (1, 6), generator=g).to(DEV)
    table = ops.MaskTable.causal(1, 70, DEV)
    outs = {}
    with _lib.use_lab(0) as lab, torch.no_grad():
        for fault in (False, True):
            cache = lm(inputs_embeds=x, attention_mask=table, use_cache=True, cache_capacity=96).past_key_values
            if fault:
                lab.aki_lab_set_chain_fault((1 << 8) | 4, 3)
                with pytest.warns(RuntimeWarning, match="decode chain"):
                    outs[fault] = lm(input_ids=new, past_key_values=cache).logits
                assert cache.chain is None and cache.chain_disabled and int(cache.cache_len[0]) == 76
            else:
                outs[fault] = lm(input_ids=new, past_key_values=cache).logits
                assert cache.chain is not None
        lab.aki_lab_set_chain_fault(0, 0)
    assert torch.equal(outs[True], outs[False])
