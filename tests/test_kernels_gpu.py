"""GPU parity tests: every HIP kernel, called through the C ABI (aki_amd.ops -> libaki_mi355x.so),
against the numpy oracle on the same seeded inputs.  Run with ``pytest -m gpu`` on an MI355X.

Tolerances (BASELINE.json: 1e-3 bf16 / 1e-5 fp32):
  fp32 : |hip - oracle| <= 1e-5 * max(1, max|oracle|)
  bf16 : |hip - oracle_f32(bf16-rounded inputs)| <= 1e-3 * max(1, max|oracle|) + 2^-8 * |oracle|
         (the second term is the half-ulp of the bf16 OUTPUT format, which no kernel can avoid)
Integer / index / byte outputs are bit-exact.
"""
import json

import numpy as np
import pytest
import torch

from conftest import load_golden
from golden import gen
import aki_oracle as O

pytestmark = pytest.mark.gpu

DEV = "cuda"


def _ops():
    from aki_amd import ops
    return ops


def t(x, dtype):
    return torch.from_numpy(np.ascontiguousarray(x)).to(DEV).to(dtype)


def n(x):
    return x.detach().float().cpu().numpy()


def rnd(x, dtype):
    """What the device sees after the dtype cast, as f32 numpy."""
    return O.bf16_round(x) if dtype == torch.bfloat16 else np.asarray(x, dtype=np.float32)


def check(got, want, dtype, what, scale_atol=1.0):
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    assert got.shape == want.shape, f"{what}: shape {got.shape} vs {want.shape}"
    assert np.isfinite(got).all(), f"{what}: non-finite values in the HIP output ({np.sum(~np.isfinite(got))})"
    mx = max(1.0, float(np.abs(want).max()))
    if dtype == torch.bfloat16:
        tol = 1e-3 * scale_atol * mx + 2.0 ** -8 * np.abs(want)
    else:
        tol = 1e-5 * scale_atol * mx + 0 * want
    err = np.abs(got - want)
    bad = err > tol
    from conftest import record_parity
    record_parity(what, dtype, err.max() if err.size else 0.0, err.mean() if err.size else 0.0, np.abs(want).max() if want.size else 0.0,
                  f"{'1e-3' if dtype == torch.bfloat16 else '1e-5'}*{scale_atol:g}*max(1,max|ref|)" + (" + 2^-8|ref|" if dtype == torch.bfloat16 else ""))
    if bad.any():
        idx = np.unravel_index(np.argmax(err - tol), err.shape)
        raise AssertionError(f"{what}: {bad.sum()}/{bad.size} elements out of tolerance; worst at {idx}: "
                             f"hip={got[idx]:.6g} oracle={want[idx]:.6g} err={err[idx]:.3g} tol={tol[idx]:.3g} "
                             f"(max|oracle|={mx:.3g}, mean err={err.mean():.3g})")


DTYPES = [torch.bfloat16, torch.float32]


# ------------------------------------------------------------------------------------------------
# linear
# ------------------------------------------------------------------------------------------------
@pytest.fixture(params=[0, 1, 2, 3, 5], ids=["tile-auto", "tile-256", "tile-128", "tile-128x96", "tile-64x64-ring"])
def gemm_tile(request):
    """Run the bf16 GEMM tests under the heuristic (product library) and with each tile configuration forced - the forcing
    switch exists only in the lab twin of the library (libaki_mi355x_lab.so: same sources + aki_lab_set_gemm_tile)."""
    from aki_amd import _lib
    if request.param == 0:
        _lib.load()
        yield 0
        return
    with _lib.use_lab(request.param):
        yield request.param


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,K", [(300, 512, 256), (256, 256, 64), (1, 768, 128), (700, 1152, 640), (513, 36, 192), (120, 4096, 512), (97, 40000, 256),
                                   (300, 512, 128), (300, 512, 192)])   # K = 128 / 192: two / three K-steps through the three-deep rings and the residual blocks
def test_linear_plain_bias_act_residual(dtype, M, N, K, gemm_tile):
    ops = _ops()
    if dtype == torch.float32 and gemm_tile:
        pytest.skip("tile configuration only affects the bf16 kernel")
    rng = gen.rng_for(f"lin{M}{N}{K}")
    x = rng.standard_normal((M, K), dtype=np.float32)
    w = rng.standard_normal((N, K), dtype=np.float32) * 0.05
    b = rng.standard_normal((N,), dtype=np.float32) * 0.1
    r = rng.standard_normal((M, N), dtype=np.float32)
    xr, wr, br, rr = (rnd(a, dtype) for a in (x, w, b, r))
    base = xr @ wr.T
    y = ops.linear(t(x, dtype), t(w, dtype))
    check(n(y), base, dtype, "plain")
    y = ops.linear(t(x, dtype), t(w, dtype), bias=t(b, dtype))
    check(n(y), base + br, dtype, "bias")
    y = ops.linear(t(x, dtype), t(w, dtype), bias=t(b, dtype), act=ops.ACT_GELU_ERF)
    check(n(y), O.gelu_erf((base + br).astype(np.float32)), dtype, "gelu_erf")
    y = ops.linear(t(x, dtype), t(w, dtype), bias=t(b, dtype), act=ops.ACT_GELU_TANH)
    check(n(y), O.gelu_tanh((base + br).astype(np.float32)), dtype, "gelu_tanh")
    y = ops.linear(t(x, dtype), t(w, dtype), residual=t(r, dtype))
    check(n(y), base + rr, dtype, "residual")
    mod = 7
    y = ops.linear(t(x, dtype), t(w, dtype), bias=t(b, dtype), residual=t(r[:mod], dtype), res_row_mod=mod)
    check(n(y), base + br + rr[np.arange(M) % mod], dtype, "residual row-mod (pos-emb)")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,Nout,K", [(300, 256, 128), (130, 320, 64), (515, 128, 192), (120, 2048, 512), (64, 9000, 256)])
def test_linear_swiglu(dtype, M, Nout, K, gemm_tile):
    ops = _ops()
    if dtype == torch.float32 and gemm_tile:
        pytest.skip("tile configuration only affects the bf16 kernel")
    rng = gen.rng_for(f"swiglu{M}{Nout}{K}")
    x = rng.standard_normal((M, K), dtype=np.float32)
    w = rng.standard_normal((2 * Nout, K), dtype=np.float32) * 0.08
    xr, wr = rnd(x, dtype), rnd(w, dtype)
    up = xr @ wr.T
    want = up[:, Nout:] * O.silu(up[:, :Nout])
    y = ops.linear(t(x, dtype), t(w, dtype), act=ops.ACT_SWIGLU)
    check(n(y), want, dtype, "swiglu")


def test_linear_identity_layout_bf16(gemm_tile):
    """A = I against an asymmetric W: catches a transposed or permuted accumulator map (cdna guide section 3)."""
    ops = _ops()
    K = N = 256
    M = 256
    x = np.eye(M, K, dtype=np.float32)
    w = (np.arange(N)[:, None] * 3 + np.arange(K)[None, :] % 17).astype(np.float32) % 251
    y = ops.linear(t(x, torch.bfloat16), t(w, torch.bfloat16))
    assert np.array_equal(n(y), O.bf16_round(w).T[:M])


def test_linear_full_size_bf16_rows():
    """AKI-4B gate_up shape (M=5240, K=3072, N=16384): sampled rows against the oracle + linearity property."""
    ops = _ops()
    M, K, Nout = 5240, 3072, 8192
    g = torch.Generator(device="cpu").manual_seed(0)
    x = (torch.randn(M, K, generator=g) * 1.0).to(torch.bfloat16)
    w = (torch.randn(2 * Nout, K, generator=g) * 0.02).to(torch.bfloat16)
    y = ops.linear(x.to(DEV), w.to(DEV))
    rows = [0, 1, 255, 256, 4095, 5239]
    want = x[rows].float().numpy() @ w.float().numpy().T
    check(n(y[rows]), want, torch.bfloat16, "gate_up rows")
    ys = ops.linear(x.to(DEV), w.to(DEV), act=ops.ACT_SWIGLU)
    want_s = want[:, Nout:] * O.silu(want[:, :Nout].astype(np.float32))
    check(n(ys[rows]), want_s, torch.bfloat16, "gate_up swiglu rows")
    # linearity: f(2x) == 2 f(x) exactly in bf16 (power-of-two scaling commutes with rounding)
    y2 = ops.linear((x * 2).to(DEV), w.to(DEV))
    assert torch.equal(y2, y * 2)


# ------------------------------------------------------------------------------------------------
# norms
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("rows,cols", [(5, 3072), (33, 1152), (2, 192), (1, 64)])
def test_norms(dtype, rows, cols):
    ops = _ops()
    rng = gen.rng_for(f"norm{rows}{cols}")
    x = rng.standard_normal((rows, cols), dtype=np.float32) * 3 + 0.5
    w = 1 + 0.1 * rng.standard_normal((cols,), dtype=np.float32)
    b = 0.1 * rng.standard_normal((cols,), dtype=np.float32)
    xr, wr, br = rnd(x, dtype), rnd(w, dtype), rnd(b, dtype)
    y = ops.rmsnorm(t(x, dtype), t(w, dtype), 1e-5)
    if dtype == torch.bfloat16:
        xf = xr / np.sqrt((xr * xr).mean(-1, keepdims=True) + 1e-5)
        want = wr * O.bf16_round(xf.astype(np.float32))     # HF:phi3 266-284: cast before the gain
    else:
        want = O.rms_norm(xr, wr, 1e-5)
    check(n(y), want, dtype, "rmsnorm")
    y = ops.layernorm(t(x, dtype), t(w, dtype), t(b, dtype), 1e-6)
    check(n(y), O.layer_norm(xr, wr, br, 1e-6), dtype, "layernorm")


# ------------------------------------------------------------------------------------------------
# attention core + fused op
# ------------------------------------------------------------------------------------------------
def _attn_case(tag, B, H, L, dtype, pad=True, dead=True):
    rng = gen.rng_for("attn_" + tag)
    q = rng.standard_normal((B, H, L, 96), dtype=np.float32)
    k = rng.standard_normal((B, H, L, 96), dtype=np.float32)
    v = rng.standard_normal((B, H, L, 96), dtype=np.float32)
    Nv = max(1, min(144, L // 4))
    am = np.ones((B, L), dtype=np.int64)
    seq = [L] * B
    rects = []
    for b in range(B):
        s = (3 + 5 * b) % max(1, L - Nv)
        e = L - 2 - b if b % 3 != 2 else 0      # every third sample: no <|assistant|> -> pure causal
        if pad and b % 2 == 0 and L > 8:
            am[b, L - L // 8:] = 0
        if dead and b == B - 1 and L > 16:
            seq[b] = L - L // 5                   # bottom rows are stacking padding
            am[b, seq[b]:] = 0
        rects.append([O.clamp_span(seq[b], s, s + Nv, e)])
    return q, k, v, am, seq, rects


def _oracle_core(q, k, v, am, seq, rects, dtype):
    am2 = am.copy()
    B, H, L, _ = q.shape
    qr, kr, vr = rnd(q, dtype), rnd(k, dtype), rnd(v, dtype)
    out = O.mma_attention_core_spans(qr, kr, vr, am2, rects, 96 ** -0.5)
    # rows >= seq_len are all-zero mask rows in the reference -> uniform softmax over all L columns
    for b in range(B):
        if seq[b] < L:
            mean = vr[b].mean(axis=1)                       # (H, 96)
            out[b, seq[b]:] = mean.reshape(-1)[None]
    return out


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,H,L", [(1, 1, 32), (2, 2, 40), (2, 2, 200), (3, 2, 333), (2, 4, 655), (1, 2, 1000)])
def test_mma_attn_core(dtype, B, H, L):
    ops = _ops()
    q, k, v, am, seq, rects = _attn_case(f"{B}{H}{L}", B, H, L, dtype)
    table = ops.MaskTable.from_host(rects, am, seq, DEV)
    o, lse = ops.mma_attn_core(t(q, dtype), t(k, dtype), t(v, dtype), table, 96 ** -0.5, return_lse=True)
    want = _oracle_core(q, k, v, am, seq, rects, dtype)
    check(n(o), want, dtype, f"attn core B{B} H{H} L{L}")
    # dense mask from the same table is bit-exact with the oracle's dense restatement
    dense = ops.mask_dense(table, B).cpu().numpy()
    for b in range(B):
        wantm = np.zeros((L, L), dtype=np.int64)
        wantm[:seq[b], :seq[b]] = O.mask_from_spans(am[b, :seq[b]], rects[b])[0]
        assert np.array_equal(dense[b, 0], wantm), f"dense mask sample {b}"


@pytest.mark.parametrize("dtype", DTYPES)
def test_mma_attn_core_no_table_is_causal(dtype):
    ops = _ops()
    B, H, L = 2, 2, 150
    q, k, v, _, _, _ = _attn_case("causal", B, H, L, dtype, pad=False, dead=False)
    o = ops.mma_attn_core(t(q, dtype), t(k, dtype), t(v, dtype), ops.MaskTable.causal(B, L, DEV), 96 ** -0.5)
    want = O.mma_attention_core_spans(rnd(q, dtype), rnd(k, dtype), rnd(v, dtype), np.ones((B, L), dtype=np.int64),
                                      [[(0, 0, 0, 0)]] * B, 96 ** -0.5)
    check(n(o), want, dtype, "causal")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("dead_rows", [1, 0])
def test_mma_attn_core_left_padding_dead_rows(dtype, dead_rows):
    """Left padding: the first rows are inside seq_len but see no valid column -> uniform softmax over all L columns
    (reference convention, AKI_DEAD_ROWS_UNIFORM) or zeros (AKI_DEAD_ROWS_ZERO)."""
    ops = _ops()
    B, H, L = 2, 2, 100
    q, k, v, _, _, _ = _attn_case("leftpad", B, H, L, dtype, pad=False, dead=False)
    am = np.ones((B, L), dtype=np.int64)
    am[0, :7] = 0
    am[1, :40] = 0
    rects = [[(10, 30, 30, 80)], [(0, 0, 0, 0)]]
    table = ops.MaskTable.from_host(rects, am, None, DEV)
    o = ops.mma_attn_core(t(q, dtype), t(k, dtype), t(v, dtype), table, 96 ** -0.5, dead_rows=dead_rows)
    want = O.mma_attention_core_spans(rnd(q, dtype), rnd(k, dtype), rnd(v, dtype), am, rects, 96 ** -0.5)
    if not dead_rows:
        want[0, :7] = 0
        want[1, :40] = 0
    check(n(o), want, dtype, f"left padding, dead_rows={dead_rows}")


def test_mma_attn_core_online_softmax_rescale_forced_bf16():
    """Force the running max to jump at a late KV tile (cdna guide rule 26): spike one key against every query."""
    ops = _ops()
    B, H, L = 1, 1, 256
    rng = gen.rng_for("spike")
    q = rng.standard_normal((B, H, L, 96), dtype=np.float32)
    k = rng.standard_normal((B, H, L, 96), dtype=np.float32) * 0.1
    v = rng.standard_normal((B, H, L, 96), dtype=np.float32)
    k[0, 0, 200] = q[0, 0, 230] * 4.0     # key 200 dominates row 230 (and changes others)
    k[0, 0, 70] = q[0, 0, 100] * 3.0
    table = ops.MaskTable.from_host([[(0, 0, 0, 0)]], np.ones((B, L)), None, DEV)
    o = ops.mma_attn_core(t(q, torch.bfloat16), t(k, torch.bfloat16), t(v, torch.bfloat16), table, 96 ** -0.5)
    want = O.mma_attention_core_spans(O.bf16_round(q), O.bf16_round(k), O.bf16_round(v), np.ones((B, L), dtype=np.int64),
                                      [[(0, 0, 0, 0)]], 96 ** -0.5)
    check(n(o), want, torch.bfloat16, "spiked rows", scale_atol=2.0)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("tag", ["small", "mid"])
def test_fused_mma_attn_vs_reference_golden(dtype, tag):
    """x -> fused (QKV proj + RoPE + MMA attention) -> o_proj, against Phi3Attention run by the reference harness
    (fp32 golden y32; the bf16 run is compared with the same fp32 golden under the bf16 tolerance)."""
    ops = _ops()
    g = load_golden(f"attn_block_{tag}.npz")
    shapes = [(k_, tuple(s)) for k_, s in json.loads(str(g["shapes"]))]
    p = gen.fill_params(shapes, 31)
    B, L = g["am"].shape
    d = p["qkv_proj.weight"].shape[1]
    H = d // 96
    x = gen.rng_for("attn_block_" + tag).standard_normal((B, L, d), dtype=np.float32)
    rects = [[O.clamp_span(L, *map(int, g["spans"][b]))] for b in range(B)]
    table = ops.MaskTable.from_host(rects, g["am"], None, DEV)
    cos, sin = torch.from_numpy(g["cos"][0]).to(DEV), torch.from_numpy(g["sin"][0]).to(DEV)
    o = ops.mma_attn(t(x, dtype), t(p["qkv_proj.weight"], dtype), cos, sin, table, H)
    y = ops.linear(o, t(p["o_proj.weight"], dtype))
    if dtype == torch.float32:
        check(n(y), g["y32"], dtype, "fused attention block fp32 vs reference")
    else:
        # oracle on bf16-rounded inputs, f32 arithmetic
        m4 = gen.unpack_mask_bits(g["mask_bits"], tuple(g["mask_shape"]))
        want = O.phi3_attention(O.bf16_round(x), O.bf16_round(p["qkv_proj.weight"]), O.bf16_round(p["o_proj.weight"]),
                                g["cos"], g["sin"], O.invert_mask_441(m4), H)
        # two chained bf16 roundings (attention output, then o_proj) -> 2x the single-kernel tolerance
        check(n(y), want, dtype, "fused attention block bf16 vs oracle", scale_atol=4.0)
        # and it is at least as close to the fp32 reference as the reference's own bf16 eager path
        e_hip = np.abs(n(y) - g["y32"]).mean()
        e_ref = np.abs(g["y16"] - g["y32"]).mean()
        assert e_hip <= 1.25 * e_ref + 1e-4, f"HIP bf16 mean error {e_hip:.3g} vs reference bf16 eager {e_ref:.3g}"


def test_qkv_rope_stage_bf16():
    ops = _ops()
    B, L, H = 2, 300, 4
    d = 96 * H
    rng = gen.rng_for("qkvrope")
    x = rng.standard_normal((B, L, d), dtype=np.float32)
    w = rng.standard_normal((3 * d, d), dtype=np.float32) * 0.05
    pos = np.stack([np.arange(L), np.arange(L)[::-1]]).astype(np.int64)       # per-sample position ids
    cos, sin = O.rope_cos_sin(np.arange(L)[None], 96)
    q, k, v = ops.qkv_rope(t(x, torch.bfloat16), t(w, torch.bfloat16), torch.from_numpy(cos[0]).to(DEV),
                           torch.from_numpy(sin[0]).to(DEV), H, position_ids=torch.from_numpy(pos).to(DEV))
    qkv = O.bf16_round(x) @ O.bf16_round(w).T
    hd = lambda a: a.reshape(B, L, H, 96).transpose(0, 2, 1, 3)
    cp, sp = cos[0][pos], sin[0][pos]
    qw, kw = O.apply_rope(hd(qkv[..., :d]), hd(qkv[..., d:2 * d]), cp, sp)
    check(n(q), qw, torch.bfloat16, "q rope")
    check(n(k), kw, torch.bfloat16, "k rope")
    check(n(v), hd(qkv[..., 2 * d:]), torch.bfloat16, "v")


def test_qkv_rope_full_width_with_m_tail_split_bf16():
    """AKI-4B width (d=3072, 32 heads) at M = 1380 = 5*256 + 100: exercises the big-tile + M-tail launch pair of the
    QKV kernel (token index / position ids must stay global in the tail launch)."""
    ops = _ops()
    B, L, H = 2, 690, 32
    d = 96 * H
    g = torch.Generator(device="cpu").manual_seed(5)
    x = torch.randn(B, L, d, generator=g).to(torch.bfloat16)
    w = (torch.randn(3 * d, d, generator=g) * 0.02).to(torch.bfloat16)
    cos, sin = O.rope_cos_sin(np.arange(L)[None], 96)
    q, k, v = ops.qkv_rope(x.to(DEV), w.to(DEV), torch.from_numpy(cos[0]).to(DEV), torch.from_numpy(sin[0]).to(DEV), H)
    qkv = x.float().numpy().reshape(B * L, d) @ w.float().numpy().T
    qkv = qkv.reshape(B, L, 3 * d)
    hd = lambda a: a.reshape(B, L, H, 96).transpose(0, 2, 1, 3)
    qw, kw = O.apply_rope(hd(qkv[..., :d]), hd(qkv[..., d:2 * d]), cos, sin)
    check(n(q), qw, torch.bfloat16, "q rope (tail split)")
    check(n(k), kw, torch.bfloat16, "k rope (tail split)")
    check(n(v), hd(qkv[..., 2 * d:]), torch.bfloat16, "v (tail split)")


def test_linear_m_tail_split_with_row_mod_residual_bf16():
    """Split launch + residual row-modulo (position-embedding style): the modulo must use the global row index."""
    ops = _ops()
    M, N, K, mod = 1380, 16384, 256, 729
    g = torch.Generator(device="cpu").manual_seed(6)
    x = torch.randn(M, K, generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g) * 0.05).to(torch.bfloat16)
    r = torch.randn(mod, N, generator=g).to(torch.bfloat16)
    b = (torch.randn(N, generator=g) * 0.1).to(torch.bfloat16)
    y = ops.linear(x.to(DEV), w.to(DEV), bias=b.to(DEV), residual=r.to(DEV), res_row_mod=mod)
    want = x.float().numpy() @ w.float().numpy().T + b.float().numpy() + r.float().numpy()[np.arange(M) % mod]
    check(n(y), want, torch.bfloat16, "tail split + row-mod residual")
    y2 = ops.linear(x.to(DEV), w.to(DEV), residual=y)
    check(n(y2), x.float().numpy() @ w.float().numpy().T + n(y), torch.bfloat16, "tail split + plain residual")


def test_attention_full_size_bf16_vs_f32_kernel_and_oracle_heads():
    """Config 2 of BASELINE.json (B=8, H=32, L=655): bf16 MFMA kernel vs the exact-f32 kernel on all heads, and BOTH kernels
    directly vs the numpy oracle on eight (batch, head) pairs - one per sample, spread over the heads; plus the convexity
    property of softmax(V)."""
    ops = _ops()
    B, H, L = 8, 32, 655
    g = torch.Generator(device="cpu").manual_seed(1)
    q = torch.randn(B, H, L, 96, generator=g)
    k = torch.randn(B, H, L, 96, generator=g)
    v = torch.randn(B, H, L, 96, generator=g)
    am = np.ones((B, L), dtype=np.int64)
    lens = [655, 600, 655, 512, 655, 640, 655, 655]
    rects = []
    for b in range(B):
        am[b, lens[b]:] = 0
        rects.append([O.clamp_span(lens[b], 6, 150, 638 if b != 3 else 0)])
    table = ops.MaskTable.from_host(rects, am, lens, DEV)
    qb, kb, vb = (a.to(torch.bfloat16).to(DEV) for a in (q, k, v))
    o16 = ops.mma_attn_core(qb, kb, vb, table, 96 ** -0.5)
    o32 = ops.mma_attn_core(qb.float(), kb.float(), vb.float(), table, 96 ** -0.5)
    check(n(o16), n(o32), torch.bfloat16, "bf16 MFMA kernel vs exact-f32 kernel, all heads")
    for (b, h) in [(0, 0), (1, 5), (2, 31), (3, 17), (4, 8), (5, 23), (6, 12), (7, 30)]:
        sl = lambda a: a[b:b + 1, h:h + 1].to(torch.bfloat16).float().numpy()
        want = O.mma_attention_core_spans(sl(q), sl(k), sl(v), am[b:b + 1], [rects[b]], 96 ** -0.5)
        got = n(o32)[b, :lens[b], h * 96:(h + 1) * 96]
        check(got, want[0, :lens[b]], torch.float32, f"f32 kernel vs oracle (b={b}, h={h})", scale_atol=2.0)
        got16 = n(o16)[b, :lens[b], h * 96:(h + 1) * 96]
        check(got16, want[0, :lens[b]], torch.bfloat16, f"bf16 MFMA kernel vs oracle at the benchmark shape (b={b}, h={h})", scale_atol=2.0)
    # property: rows are convex combinations of V rows -> every output lies inside [min V, max V] per channel
    vmin = vb.float().amin(dim=2)      # B,H,96
    vmax = vb.float().amax(dim=2)
    o = o16.float().reshape(B, L, H, 96).permute(0, 2, 1, 3)
    assert bool(((o >= vmin[:, :, None] - 2e-2) & (o <= vmax[:, :, None] + 2e-2)).all())


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,H,Lq,Lk,Dh", [(2, 16, 576, 576, 72), (3, 8, 144, 873, 64), (2, 2, 16, 16, 32), (1, 4, 100, 333, 96),
                                          (2, 16, 729, 729, 72)])
def test_plain_attention_vision_side(dtype, B, H, Lq, Lk, Dh):
    """aki_attn_fwd (SigLIP 16x72, Perceiver 8x64 with 144 queries over 873 keys) on strided views of fused projections."""
    ops = _ops()
    rng = gen.rng_for(f"ncattn{B}{H}{Lq}{Lk}{Dh}")
    if Lq == Lk:   # SigLIP style: one fused [B, L, 3, H, Dh] projection output
        qkv = rng.standard_normal((B, Lq, 3, H, Dh), dtype=np.float32)
        tq = t(qkv, dtype)
        q_, k_, v_ = tq[:, :, 0], tq[:, :, 1], tq[:, :, 2]
        qn, kn, vn = (rnd(qkv[:, :, i], dtype) for i in range(3))
    else:          # Perceiver style: q from the latents, k/v from one fused [B, Lk, 2, H, Dh] projection
        q = rng.standard_normal((B, Lq, H, Dh), dtype=np.float32)
        kv = rng.standard_normal((B, Lk, 2, H, Dh), dtype=np.float32)
        tkv = t(kv, dtype)
        q_, k_, v_ = t(q, dtype), tkv[:, :, 0], tkv[:, :, 1]
        qn, kn, vn = rnd(q, dtype), rnd(kv[:, :, 0], dtype), rnd(kv[:, :, 1], dtype)
    o = ops.attention(q_, k_, v_, Dh ** -0.5)
    s = np.einsum("bqhd,bkhd->bhqk", qn, kn) * np.float32(Dh ** -0.5)
    want = np.einsum("bhqk,bkhd->bqhd", O.softmax(s, -1), vn).reshape(B, Lq, H * Dh)
    # P = softmax(...) is rounded to bf16 before P.V exactly like the reference's eager path (`.to(query.dtype)`,
    # HF:siglip 241): each weight carries 2^-9 relative error, which does not average out over a handful of keys,
    # so the absolute term is 2e-3 here.
    check(n(o), want, dtype, f"plain attention Dh={Dh} Lq={Lq} Lk={Lk}", scale_atol=2.0)


# ------------------------------------------------------------------------------------------------
# splice + dense mask vs the reference's own outputs
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", DTYPES)
def test_splice_vs_reference_golden(dtype):
    ops = _ops()
    g = load_golden("tiny_e2e.npz")
    T = gen.TINY
    shapes = [(k_, tuple(s)) for k_, s in json.loads(str(g["shapes"]))]
    p = gen.fill_params(shapes, 11)
    W = p["lang_model.model.embed_tokens.weight"]
    Wadd = p["lang_model.model.embed_tokens.additional_embedding.weight"]
    emb, labels, table, plan = ops.splice(torch.from_numpy(g["lang_x"]).to(DEV), torch.from_numpy(g["attention_mask"]).to(DEV),
                                          torch.from_numpy(g["labels"]).to(DEV), t(W, dtype), t(Wadd, dtype), T["vocab"] - 1,
                                          t(g["vision_tokens"], dtype), T["media_token_id"], T["pad_token_id"])
    want = rnd(g["inputs_embeds"], dtype)
    if dtype == torch.float32:
        assert np.array_equal(n(emb), want), "inputs_embeds must be a bit-exact gather/copy"
    else:
        # the vision tokens were rounded on the way in; gather/copy itself is exact
        assert np.array_equal(n(emb), want)
    assert np.array_equal(labels.cpu().numpy(), g["new_labels"])
    dense = ops.mask_dense(table, g["lang_x"].shape[0]).cpu().numpy()
    assert np.array_equal(dense, gen.unpack_mask_bits(g["mask_bits"], tuple(g["mask_shape"])))
    # left padding (generate path, src/aki.py:171)
    gl = load_golden("tiny_splice_left.npz")
    emb_l, _, table_l, _ = ops.splice(torch.from_numpy(g["lang_x"]).to(DEV), torch.from_numpy(g["attention_mask"]).to(DEV), None,
                                      t(W, dtype), t(Wadd, dtype), T["vocab"] - 1, t(g["vision_tokens"], dtype),
                                      T["media_token_id"], T["pad_token_id"], padding_side="left")
    assert np.array_equal(n(emb_l), rnd(gl["inputs_embeds"], dtype))
    assert np.array_equal(ops.mask_dense(table_l, g["lang_x"].shape[0]).cpu().numpy(),
                          gen.unpack_mask_bits(gl["mask_bits"], tuple(gl["mask_shape"])))


def test_mask_dense_all_reference_cases():
    ops = _ops()
    g = load_golden("mask_cases.npz")
    for i, (am, s, tt, e) in enumerate(gen.mask_cases()):
        nn = len(am)
        table = ops.MaskTable.from_host([[O.clamp_span(nn, s, tt, e)]], am[None], None, DEV)
        dense = ops.mask_dense(table, 1).cpu().numpy()
        assert np.array_equal(dense[0], gen.unpack_mask_bits(g[f"bits_{i}"], (1, nn, nn))), f"mask case {i}"


def test_mask_to_table_all_reference_cases():
    """The reference's hand-off type (dense (B,1,L,L) int64 0/1, src/vlm.py:589-603) -> table -> dense is the identity
    on every mask the reference itself generated (31 golden cases), alone and stacked into padded batches the way
    stack_with_padding_2D_attention does (zero rows / columns at the bottom / right)."""
    ops = _ops()
    g = load_golden("mask_cases.npz")
    cases = gen.mask_cases()
    dense = [gen.unpack_mask_bits(g[f"bits_{i}"], (1, len(c[0]), len(c[0])))[0].astype(np.int64) for i, c in enumerate(cases)]
    for i, d in enumerate(dense):
        m = torch.from_numpy(d)[None, None].to(DEV)
        table = ops.mask_to_table(m)
        assert torch.equal(ops.mask_dense(table, 1), m), f"mask case {i}"
        rects, valid, seq_len = O.mask_to_table(d)                      # the oracle's restatement of the conversion
        assert [r for r in table.rects[0].tolist() if r != [0, 0, 0, 0]] == [list(r) for r in rects], f"mask case {i}"
        assert int(table.seq_lens[0]) == seq_len
        want_bits = ops.MaskTable.from_host([[]], valid[None], None, DEV).col_valid_bits
        assert torch.equal(table.col_valid_bits, want_bits), f"mask case {i}"
    Lmax = max(d.shape[0] for d in dense)
    stacked = np.zeros((len(dense), 1, Lmax, Lmax), dtype=np.int64)
    for i, d in enumerate(dense):
        stacked[i, 0, :d.shape[0], :d.shape[0]] = d
    m = torch.from_numpy(stacked).to(DEV)
    table = ops.mask_to_table(m, max_rects=1)              # the reference's masks need exactly one rectangle
    assert torch.equal(ops.mask_dense(table, len(dense)), m)
    # other dtypes a caller may hold the same mask in
    for dt in (torch.bool, torch.int32, torch.float32, torch.bfloat16):
        assert torch.equal(ops.mask_dense(ops.mask_to_table(m.to(dt)), len(dense)), m)


def test_mask_to_table_multi_image_and_rejections():
    ops = _ops()
    L = 200
    am = np.ones((2, L), dtype=bool)
    am[1, 150:] = False
    rects = [[(4, 20, 20, 120), (30, 46, 46, 120), (60, 76, 76, 120)], [(0, 16, 16, 90)]]
    t0 = ops.MaskTable.from_host(rects, am, [L, 170], DEV)
    dense = ops.mask_dense(t0, 2)
    t1 = ops.mask_to_table(dense)
    assert torch.equal(ops.mask_dense(t1, 2), dense)
    assert t1.rects[0, :3].tolist() == [list(r) for r in rects[0]] and t1.seq_lens.tolist() == [L, 170]
    # a rectangle that dips below the diagonal (col_lo < row_hi) is still one rectangle
    t2 = ops.MaskTable.from_host([[(10, 40, 20, 100)]], np.ones((1, L), dtype=bool), None, DEV)
    d2 = ops.mask_dense(t2, 1)
    assert torch.equal(ops.mask_dense(ops.mask_to_table(d2, max_rects=1), 1), d2)
    # outside the family: a hole in the causal triangle, a two-interval row, too many row groups
    bad = dense.clone(); bad[0, 0, 100, 50] = 0
    with pytest.raises(ops.AkiError):
        ops.mask_to_table(bad)
    bad = dense.clone(); bad[0, 0, 10, 130:140] = 1
    with pytest.raises(ops.AkiError):
        ops.mask_to_table(bad)
    with pytest.raises(ops.AkiError):
        ops.mask_to_table(dense, max_rects=2)
    full = torch.ones((1, 1, 64, 64), dtype=torch.int64, device=DEV)      # bidirectional = one rectangle under the diagonal rule
    tf = ops.mask_to_table(full, max_rects=1)
    assert tf.rects[0, 0].tolist() == [0, 63, 1, 64] and torch.equal(ops.mask_dense(tf, 1), full)
    stair = torch.tril(torch.ones((64, 64), dtype=torch.int64, device=DEV))
    for r in range(0, 30, 2):
        stair[r, r + 1:r + 3 + r] = 1                                       # one row group per row: 15 > 8
    with pytest.raises(ops.AkiError):
        ops.mask_to_table(stair[None, None])


def test_integration_md_binding_runs_verbatim(monkeypatch):
    """INTEGRATION.md section B, executed as written: the dense mask of the reference -> table, then the fused MMA op
    through the document's own ctypes structure; must equal aki_amd.ops on the same inputs bit for bit."""
    from test_abi_cpu import integration_md_binding
    from aki_amd import _lib
    from types import SimpleNamespace
    ops = _ops()
    _lib.load()
    monkeypatch.setenv("AKI_MI355X_SO", _lib.LIB_PATH)
    ns = {}
    exec(compile(integration_md_binding(), "INTEGRATION.md#B", "exec"), ns)
    rng = gen.rng_for("integration_md")
    B, L, H, Dh = 2, 150, 4, 96
    d = H * Dh
    am = np.ones((B, L), dtype=bool); am[1, 130:] = False
    t0 = ops.MaskTable.from_host([[(5, 21, 21, 100)], [(3, 19, 19, 80)]], am, [L, 140], DEV)
    dense = ops.mask_dense(t0, B)
    x = t(rng.standard_normal((B, L, d), dtype=np.float32), torch.bfloat16)
    wqkv = t(rng.standard_normal((3 * d, d), dtype=np.float32) * 0.05, torch.bfloat16)
    wo = t(rng.standard_normal((d, d), dtype=np.float32) * 0.05, torch.bfloat16)
    cos, sin = O.rope_cos_sin(np.arange(L)[None], Dh)
    tc, ts = torch.from_numpy(cos).to(DEV), torch.from_numpy(sin).to(DEV)
    module = SimpleNamespace(config=SimpleNamespace(num_attention_heads=H), head_dim=Dh, qkv_proj=SimpleNamespace(weight=wqkv),
                             o_proj=lambda o: ops.linear(o, wo))
    rects, bits, seq_lens = ns["mask_table"](dense)
    got = ns["mma_attention"](module, x, tc, ts, rects, bits, seq_lens)
    want = ops.linear(ops.mma_attn(x, wqkv, tc[0], ts[0], t0, H), wo)
    assert torch.equal(got, want)
    holed = dense.clone()
    holed[0, 0, 60, 30] = 0                       # a hole inside the causal triangle: not an MMA mask
    with pytest.raises(ValueError):
        ns["mask_table"](holed)


def test_splice_multi_image_build_defined():
    """Two images in one sample: the reference raises; the build-defined rule is checked against the
    oracle's span restatement only (parity unpinned)."""
    ops = _ops()
    T = gen.TINY
    rng = gen.rng_for("multi")
    d, Nv = 64, 8
    IMG = T["media_token_id"]
    lang_x = np.array([[1, IMG, 5, 6, IMG, 7, 8, 32001, 9, 2]], dtype=np.int64)
    W = rng.standard_normal((T["vocab"], d), dtype=np.float32)
    Wadd = rng.standard_normal((2, d), dtype=np.float32)
    vt = rng.standard_normal((1, 2, Nv, d), dtype=np.float32)
    emb, _, table, plan = ops.splice(torch.from_numpy(lang_x).to(DEV), None, None, t(W, torch.float32), t(Wadd, torch.float32),
                                     T["vocab"] - 1, t(vt, torch.float32), IMG, T["pad_token_id"], max_rects=2)
    L = 10 - 2 + 2 * Nv
    assert emb.shape == (1, L, d)
    e = n(emb)[0]
    assert np.array_equal(e[1:1 + Nv], vt[0, 0]) and np.array_equal(e[Nv + 3:2 * Nv + 3], vt[0, 1])
    assert np.array_equal(e[0], W[1]) and np.array_equal(e[-1], W[2])
    r = table.rects.cpu().numpy()[0]
    q_exp = 7 + 2 * (Nv - 1)
    assert r.tolist() == [[1, 1 + Nv, 1 + Nv, q_exp + 1], [Nv + 3, 2 * Nv + 3, 2 * Nv + 3, q_exp + 1]]


# ------------------------------------------------------------------------------------------------
# patch embed + connector
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", DTYPES)
def test_patch_embed_vs_reference_golden(dtype):
    ops = _ops()
    g = load_golden("patch_embed_full.npz")
    shapes = [(k_, tuple(s)) for k_, s in json.loads(str(g["shapes"]))]
    p = gen.fill_params(shapes, 41)
    x = gen.rng_for("patch_embed").random((2, 3, 384, 384), dtype=np.float32) * 2 - 1
    w = ops.pad_k(t(p["patch_embedding.weight"].reshape(1152, -1), dtype))
    y = ops.patch_embed(t(x, dtype), w, t(p["patch_embedding.bias"], dtype), t(p["position_embedding.weight"], dtype), 14)
    assert y.shape == (2, 729, 1152)
    if dtype == torch.float32:
        check(n(y)[:, g["rows"]], g["y_rows"], dtype, "patch embed fp32 vs reference")
    else:
        want = O.siglip_patch_embed(O.bf16_round(x), O.bf16_round(p["patch_embedding.weight"]),
                                    O.bf16_round(p["patch_embedding.bias"]), O.bf16_round(p["position_embedding.weight"]))
        check(n(y), want, dtype, "patch embed bf16 vs oracle")


@pytest.mark.parametrize("dtype", DTYPES)
def test_connector_mlp_and_projection(dtype):
    ops = _ops()
    rows, d, di, dout = 288, 1152, 4608, 3072
    rng = gen.rng_for("connector")
    x = rng.standard_normal((rows, d), dtype=np.float32)
    lw = 1 + 0.1 * rng.standard_normal((d,), dtype=np.float32)
    lb = 0.05 * rng.standard_normal((d,), dtype=np.float32)
    w1 = rng.standard_normal((di, d), dtype=np.float32) * 0.03
    w2 = rng.standard_normal((d, di), dtype=np.float32) * 0.03
    wp = rng.standard_normal((dout, d), dtype=np.float32) * 0.03
    bp = rng.standard_normal((dout,), dtype=np.float32) * 0.05
    r = lambda a: rnd(a, dtype)
    y = ops.connector_mlp(t(x, dtype), t(lw, dtype), t(lb, dtype), t(w1, dtype), t(w2, dtype))
    want = r(x) + O.feed_forward(r(x), r(lw), r(lb), r(w1), r(w2))
    check(n(y), want, dtype, "connector mlp", scale_atol=4.0)
    y = ops.connector_proj(t(x, dtype), t(lw, dtype), t(lb, dtype), t(wp, dtype), t(bp, dtype))
    want = O.linear(O.layer_norm(r(x), r(lw), r(lb)), r(wp), r(bp))
    check(n(y), want, dtype, "connector projection", scale_atol=2.0)


def test_errors_are_loud():
    ops = _ops()
    from aki_amd._lib import AkiError
    with pytest.raises(AkiError):
        ops.linear(torch.zeros(4, 64), torch.zeros(8, 64))                      # CPU tensors: no fallback
    with pytest.raises(AkiError):
        ops.linear(torch.zeros(4, 100, device=DEV, dtype=torch.bfloat16), torch.zeros(8, 100, device=DEV, dtype=torch.bfloat16))
    with pytest.raises(AkiError):
        ops.mma_attn_core(*(torch.zeros(1, 1, 8, 80, device=DEV, dtype=torch.bfloat16),) * 3, ops.MaskTable.causal(1, 8, DEV), 0.1)


def test_config4_long_context_four_images():
    """BASELINE configs[3]: seq = 4096 with 4 interleaved images (multi-image MMA mask, build-defined rule: every image's
    rows see the columns from its own end up to <|assistant|>).  bf16 MFMA kernel vs the exact-f32 kernel on all 32 heads,
    f32 kernel vs the numpy oracle on one head, the dense mask bit-exact, and the convex-combination property."""
    ops = _ops()
    B, H, L, Nv = 1, 32, 4096, 144
    starts = [6, 900, 1800, 2700]
    q_end = L - 64
    g = torch.Generator(device="cpu").manual_seed(4)
    q, k, v = (torch.randn(B, H, L, 96, generator=g) for _ in range(3))
    am = np.ones((B, L), dtype=np.int64)
    rects = [[O.clamp_span(L, s, s + Nv, q_end) for s in starts]]
    table = ops.MaskTable.from_host(rects, am, [L], DEV)
    qb, kb, vb = (a.to(torch.bfloat16).to(DEV) for a in (q, k, v))
    o16 = ops.mma_attn_core(qb, kb, vb, table, 96 ** -0.5)
    o32 = ops.mma_attn_core(qb.float(), kb.float(), vb.float(), table, 96 ** -0.5)
    check(n(o16), n(o32), torch.bfloat16, "L=4096, 4 images: bf16 MFMA kernel vs exact-f32 kernel")
    h = 13
    sl = lambda a: a[:, h:h + 1].to(torch.bfloat16).float().numpy()
    want = O.mma_attention_core_spans(sl(q), sl(k), sl(v), am, rects, 96 ** -0.5)
    check(n(o32)[:, :, h * 96:(h + 1) * 96], want, torch.float32, "f32 kernel vs oracle, head 13", scale_atol=2.0)
    dense = ops.mask_dense(table, B).cpu().numpy()
    assert np.array_equal(dense[0, 0], O.mask_from_spans(am[0], rects[0]).reshape(L, L))
    # image rows really see their unlocked text: row starts[1] + 3 attends beyond itself, row 5 (before the first image) does not
    m = dense[0, 0]
    assert m[starts[1] + 3, starts[1] + Nv + 10] == 1 and m[starts[1] + 3, starts[1] + 5] == 0 and m[5, 6:].sum() == 0
    vmin, vmax = vb.float().amin(dim=2), vb.float().amax(dim=2)
    o = o16.float().reshape(B, L, H, 96).permute(0, 2, 1, 3)
    assert bool(((o >= vmin[:, :, None] - 2e-2) & (o <= vmax[:, :, None] + 2e-2)).all())


@pytest.mark.parametrize("seed", list(range(12)))
def test_mma_attn_core_random_masks(seed):
    """Seeded random mask configurations: 1-4 rectangles per sample anywhere in the sequence (overlapping column ranges,
    rows before or after their columns), ragged lengths, holes in the 1-D mask (left padding and interior), both dead-row
    conventions.  Exact-f32 kernel vs the numpy oracle, bf16 MFMA kernel vs the f32 kernel."""
    ops = _ops()
    rng = np.random.Generator(np.random.PCG64(1000 + seed))
    B, H = int(rng.integers(1, 4)), int(rng.integers(1, 4))
    L = int(rng.choice([33, 64, 65, 130, 257, 400, 641]))
    q, k, v = (rng.standard_normal((B, H, L, 96), dtype=np.float32) for _ in range(3))
    am = np.ones((B, L), dtype=np.int64)
    seq, rects = [], []
    nrect = int(rng.integers(1, 5))
    for b in range(B):
        nlen = L if rng.random() < 0.5 else int(rng.integers(L // 2 + 1, L + 1))
        seq.append(nlen)
        am[b, nlen:] = 0
        if rng.random() < 0.4:
            am[b, : int(rng.integers(1, max(2, nlen // 4)))] = 0          # left padding -> dead rows at the top
        if rng.random() < 0.4:
            lo = int(rng.integers(0, nlen))
            am[b, lo: min(nlen, lo + int(rng.integers(1, 9)))] = 0          # an interior hole
        rs, used = [], 0
        for _ in range(nrect):                                            # disjoint row ranges (one rectangle per row)
            if used >= nlen - 2:
                rs.append((0, 0, 0, 0))
                continue
            r0 = int(rng.integers(used, nlen - 1))
            r1 = int(rng.integers(r0 + 1, min(nlen, r0 + 1 + max(2, nlen // 3)) + 1))
            c0 = int(rng.integers(0, nlen))
            c1 = int(rng.integers(c0, nlen + 1))
            rs.append((r0, r1, c0, c1))
            used = r1
        rects.append(rs)
    dead = ops.DEAD_ROWS_UNIFORM if seed % 2 == 0 else ops.DEAD_ROWS_ZERO
    table = ops.MaskTable.from_host(rects, am, seq, DEV)
    tq, tk, tv = (torch.from_numpy(a).to(DEV) for a in (q, k, v))
    o32 = ops.mma_attn_core(tq, tk, tv, table, 96 ** -0.5, dead_rows=dead)
    o16 = ops.mma_attn_core(tq.to(torch.bfloat16), tk.to(torch.bfloat16), tv.to(torch.bfloat16), table, 96 ** -0.5, dead_rows=dead)
    # dense visibility from the table definition: valid(c) and r < seq_len and (c <= r or (r, c) inside a rectangle)
    rr, cc = np.arange(L)[:, None], np.arange(L)[None, :]
    want = np.zeros((B, L, H * 96), dtype=np.float32)
    for b in range(B):
        vis = (cc <= rr)
        for (r0, r1, c0, c1) in rects[b]:
            vis = vis | ((rr >= r0) & (rr < r1) & (cc >= c0) & (cc < c1))
        vis = vis & (am[b][None, :] != 0) & (rr < seq[b])
        for h_ in range(H):
            s_ = (q[b, h_] @ k[b, h_].T) * np.float32(96 ** -0.5)
            s_ = np.where(vis, s_, -np.inf)
            alive = vis.any(-1)
            p_ = np.zeros_like(s_)
            p_[alive] = O.softmax(s_[alive], -1)
            out = p_ @ v[b, h_]
            if dead == ops.DEAD_ROWS_UNIFORM:
                out[~alive] = v[b, h_].mean(0)                            # finfo.min convention: uniform over all L columns
            want[b, :, h_ * 96:(h_ + 1) * 96] = out
    check(n(o32), want, torch.float32, f"f32 kernel vs dense oracle (seed {seed}, B{B} H{H} L{L})", scale_atol=2.0)
    check(n(o16), rnd(n(o32), torch.float32), torch.bfloat16, f"bf16 kernel vs f32 kernel (seed {seed})", scale_atol=4.0)


def _f32_reference_rows(ops, q, k, v, table, rows):
    """Exact-f32 kernel output restricted to a few query rows (the f32 kernel is itself pinned to the numpy oracle)."""
    o32 = ops.mma_attn_core(q.float(), k.float(), v.float(), table, 96 ** -0.5)
    return o32[:, rows]


@pytest.mark.parametrize("L", [2048, 2080])
def test_mma_attn_core_block_ranking_boundary(L):
    """64 blocks of 32 rows is the last length whose blocks are ranked by extent inside the kernel; 65 blocks run in
    position order.  Both sides of the switch, with a rectangle that makes early rows walk the whole sequence."""
    ops = _ops()
    B, H = 1, 2
    g = torch.Generator(device=DEV).manual_seed(L)
    q, k, v = (torch.randn(B, H, L, 96, device=DEV, generator=g).to(torch.bfloat16) for _ in range(3))
    rects = [[(70, 214, 214, L - 40)]]
    table = ops.MaskTable.from_host(rects, np.ones((B, L)), [L] * B, DEV)
    o = ops.mma_attn_core(q, k, v, table, 96 ** -0.5)
    rows = torch.tensor([0, 31, 69, 70, 100, 213, 214, 215, 1023, 1024, L - 41, L - 40, L - 1], device=DEV)
    want = _f32_reference_rows(ops, q, k, v, table, rows)
    check(n(o[:, rows]), n(want), torch.bfloat16, f"ranking boundary L={L}", scale_atol=4.0)


def test_mma_attn_core_is_run_to_run_deterministic():
    """Race screen at the benchmark shape (1536 workgroups, two per CU): six launches on the same inputs must agree
    bit for bit, outputs and log-sum-exp.  (A VALU->MFMA operand hazard behind inline asm once made ~0.5 % of the
    outputs wobble by an ulp without ever failing a tolerance check.)"""
    ops = _ops()
    B, H, L = 8, 32, 655
    g = torch.Generator(device=DEV).manual_seed(3)
    q, k, v = (torch.randn(B, H, L, 96, device=DEV, generator=g).to(torch.bfloat16) for _ in range(3))
    am = np.ones((B, L))
    am[1, L - 50:] = 0
    table = ops.MaskTable.from_host([[(6, 150, 150, L - 17)]] * B, am, [L] * (B - 1) + [L - 90], DEV)
    o0, l0 = ops.mma_attn_core(q, k, v, table, 96 ** -0.5, return_lse=True)
    o0, l0 = o0.clone(), l0.clone()
    for i in range(5):
        o, lse = ops.mma_attn_core(q, k, v, table, 96 ** -0.5, return_lse=True)
        assert torch.equal(o, o0), f"launch {i + 1}: {int((o != o0).sum())} output elements differ from launch 0"
        assert torch.equal(lse, l0), f"launch {i + 1}: lse differs"


def test_plain_attention_is_run_to_run_deterministic():
    """Same race screen for the un-masked attention kernel (SigLIP 16 x 72 heads, 8 images of 576 patches)."""
    ops = _ops()
    g = torch.Generator(device=DEV).manual_seed(5)
    q, k, v = (torch.randn(8, 576, 16, 72, device=DEV, generator=g).to(torch.bfloat16) for _ in range(3))   # [B,L,H,Dh]
    o0 = ops.attention(q, k, v, 72 ** -0.5).clone()
    for i in range(5):
        o = ops.attention(q, k, v, 72 ** -0.5)
        assert torch.equal(o, o0), f"launch {i + 1}: {int((o != o0).sum())} elements differ"


def test_mma_attn_core_dispatch_grouping_is_invisible():
    """K+V above 128 MB switches the workgroup order from whole-grid rank-major to groups of (batch, head) pairs; the
    order must not change a single bit of any row.  B*H = 72 pairs of L = 2560 (141 MB) against the same pairs run
    24 at a time (47 MB: whole-grid order); 72 is not a multiple of the group size, so the last group is short."""
    ops = _ops()
    B, H, L = 3, 24, 2560
    g = torch.Generator(device=DEV).manual_seed(7)
    q, k, v = (torch.randn(B, H, L, 96, device=DEV, generator=g).to(torch.bfloat16) for _ in range(3))
    rects = [[(10, 154, 154, L - 8)], [(0, 0, 0, 0)], [(300, 444, 444, 2000), (900, 1044, 1044, 2000)]]
    am = np.ones((B, L))
    am[1, L - 100:] = 0
    table = ops.MaskTable.from_host(rects, am, [L, L, L - 37], DEV)
    o = ops.mma_attn_core(q, k, v, table, 96 ** -0.5)
    for b in range(B):
        tb = ops.MaskTable.from_host([rects[b]], am[b:b + 1], [[L, L, L - 37][b]], DEV)
        ob = ops.mma_attn_core(q[b:b + 1].contiguous(), k[b:b + 1].contiguous(), v[b:b + 1].contiguous(), tb, 96 ** -0.5)
        assert torch.equal(o[b:b + 1], ob), f"sample {b} differs between grouped and whole-grid dispatch"


def test_mma_attn_core_rowwise_tiles_and_odd_pair_count():
    """Image rows and text rows in one 32-row block, rectangle columns covering whole 64-key tiles (the ROWWISE bias
    path), two rectangles with different column ranges touching the same block (falls back to the visibility word), and a
    (batch, head) count that is neither a multiple of 8 nor of the 4 waves."""
    ops = _ops()
    B, H, L = 3, 3, 450
    g = torch.Generator(device=DEV).manual_seed(11)
    q, k, v = (torch.randn(B, H, L, 96, device=DEV, generator=g).to(torch.bfloat16) for _ in range(3))
    rects = [[(6, 150, 150, 440)],                       # rows 128..149 share block 4 with text rows 150..159
             [(40, 50, 128, 384), (50, 60, 192, 448)],    # two rectangles inside block 1, different columns
             [(0, 0, 0, 0)]]
    table = ops.MaskTable.from_host(rects, np.ones((B, L)), [L] * B, DEV)
    o = ops.mma_attn_core(q, k, v, table, 96 ** -0.5)
    rows = torch.arange(0, L, device=DEV)
    want = _f32_reference_rows(ops, q, k, v, table, rows)
    check(n(o), n(want), torch.bfloat16, "rowwise tiles", scale_atol=4.0)


@pytest.mark.parametrize("seed", list(range(10)))
def test_splice_random_prompts_vs_oracle(seed):
    """Seeded random prompt batches through the splice / mask-table kernels against the oracle's restatement of
    _prepare_inputs_for_forward (src/vlm.py:445-603), bit for bit: image placeholder at a random position or absent,
    <|assistant|> (32001) present or not (and before or after the image), ragged right padding of the prompts, additional
    (> max_original_id) token ids, right and left padding of the output."""
    ops = _ops()
    rng = np.random.Generator(np.random.PCG64(3000 + seed))
    B, T_, d, Nv = int(rng.integers(1, 6)), int(rng.integers(8, 80)), 64, int(rng.choice([4, 8, 144]))
    V, media, pad_id = 32011, 32011, 32000
    lang_x = rng.integers(3, 31000, size=(B, T_)).astype(np.int64)
    am = np.ones((B, T_), dtype=np.int64)
    labels = lang_x.copy()
    for b in range(B):
        nlen = T_ if rng.random() < 0.5 else int(rng.integers(T_ // 2 + 1, T_ + 1))
        am[b, nlen:] = 0
        lang_x[b, nlen:] = pad_id
        if rng.random() < 0.8:
            lang_x[b, int(rng.integers(0, nlen))] = media
        if rng.random() < 0.7:
            lang_x[b, int(rng.integers(0, nlen))] = 32001 if rng.random() < 0.9 else 32012
        labels[b] = np.where(am[b] == 0, -100, lang_x[b])
    if (lang_x == media).sum(1).max() == 0:
        lang_x[0, 1] = media
    W = rng.standard_normal((V, d), dtype=np.float32)
    Wadd = rng.standard_normal((2, d), dtype=np.float32)
    vt = rng.standard_normal((B, 1, Nv, d), dtype=np.float32)
    emb_in = O.decoupled_embedding(lang_x, W, Wadd, V - 1)
    side = "right" if seed % 2 == 0 else "left"
    want = O.prepare_inputs_for_forward(vt, lang_x, am, labels, emb_in, media, pad_id, Nv, side)
    emb, lab, table, _ = ops.splice(torch.from_numpy(lang_x).to(DEV), torch.from_numpy(am).to(DEV), torch.from_numpy(labels).to(DEV),
                                    torch.from_numpy(W).to(DEV), torch.from_numpy(Wadd).to(DEV), V - 1, torch.from_numpy(vt).to(DEV),
                                    media, pad_id, padding_side=side)
    assert np.array_equal(n(emb), want["inputs_embeds"]), "inputs_embeds must be a bit-exact gather / copy"
    assert np.array_equal(lab.cpu().numpy(), want["labels"])
    # the reference stacks the per-sample masks top-left (src/utils.py:99-108) whatever the padding side of the embeddings
    assert np.array_equal(ops.mask_dense(table, B).cpu().numpy(), want["attention_mask"])
