"""GPU parity tests of the long-sequence attention core (aki_amd/csrc/mma_attn64_bf16.hip: 64 query rows per wave, one wave per SIMD;
the product library routes L >= 1792 to it).  Three anchors:
  * its exact-maximum build (lab variant 10) must equal the 32-row kernel BIT FOR BIT, outputs and log-sum-exp, on every mask class -
    the 32-row kernel is itself pinned to the numpy oracle and the reference's golden vectors (tests/test_kernels_gpu.py);
  * the shipped build (blind softmax against the reference maximum of a rank's first tile, row sums on the matrix pipe, one
    verification per rank) against the exact-f32 kernel under the suite's bf16 bar;
  * the rare paths (cdna guide rule 26: data-dependent branches get inputs that force them): scores that outgrow the first tile's
    maximum by less than the verification bound (P up to 2^45: stays blind), by more (the row sums overflow: the rank must be walked
    again through the exact path), the build that sends EVERY tile through the exact path (lab variant 164), and the exact build
    must agree to rounding.
Reference semantics: HF:models/phi3/modeling_phi3.py:145-167 under the mask of src/vlm.py:410-443."""
import numpy as np
import pytest
import torch

import aki_oracle as O
from test_kernels_gpu import check, n, _ops, DEV

pytestmark = pytest.mark.gpu

IMG4 = [(6, 150, 150, 4032), (900, 1044, 1044, 4032), (1800, 1944, 1944, 4032), (2700, 2844, 2844, 4032)]


def _qkv(seed, B, H, L, spike=False):
    g = torch.Generator(device=DEV).manual_seed(seed)
    q, k, v = (torch.randn(B, H, L, 96, device=DEV, generator=g).to(torch.bfloat16) for _ in range(3))
    if spike:      # one key per 64-key tile that beats everything before it for one later row: the reference maximum has to be raised again and again
        for j in range(0, L, 64):
            k[:, :, j] = q[:, :, min(L - 1, j + 40)] * (0.5 + 0.4 * j / L)
    return q, k, v


def _ragged(B, L, lens):
    am = np.ones((B, L))
    for b in range(B):
        am[b, lens[b]:] = 0
    return am


CASES = {
    "causal-512": (1, 2, 512, [[(0, 0, 0, 0)]], None, None, 1),
    "one-image-655": (2, 2, 655, [[(6, 150, 150, 638)]] * 2, None, None, 1),
    "odd-207": (2, 3, 207, [[(6, 150, 150, 190)]] * 2, None, None, 1),
    "tiny-32": (1, 1, 32, [[(0, 0, 0, 0)]], None, None, 1),
    "tiny-40": (2, 2, 40, [[(3, 19, 19, 33)]] * 2, None, None, 1),
    "four-images-4096": (1, 4, 4096, [IMG4], None, None, 1),
    "position-order-4100": (1, 2, 4100, [IMG4], None, None, 1),
    "two-windows-5000": (1, 2, 5000, [[(6, 150, 150, 4900), (3000, 3144, 3144, 4900)]], None, None, 1),
    "ragged-dead-uniform": (3, 2, 1500, [[(6, 150, 150, 1400)], [(6, 150, 150, 1094)], [(6, 150, 150, 683)]], _ragged(3, 1500, [1500, 1111, 700]), [1500, 1111, 700], 1),
    "ragged-dead-zero": (3, 2, 1500, [[(6, 150, 150, 1400)], [(6, 150, 150, 1094)], [(6, 150, 150, 683)]], _ragged(3, 1500, [1500, 1111, 700]), [1500, 1111, 700], 0),
    "two-rects-one-block": (2, 3, 450, [[(40, 50, 128, 384), (50, 60, 192, 448)], [(6, 150, 150, 440)]], None, None, 1),
}


def _left_pad_case():
    B, L = 2, 1200
    am = np.ones((B, L))
    am[0, :137] = 0
    am[1, :64] = 0
    am[1, 500:520] = 0
    return (B, 2, L, [[(150, 294, 294, 1100)], [(70, 214, 214, 1150)]], am, None, 1)


CASES["left-padding-and-hole"] = _left_pad_case()


@pytest.mark.parametrize("name", list(CASES))
def test_attn64_exact_build_is_bit_identical_to_the_32_row_kernel(name):
    ops = _ops()
    from aki_amd import _lib
    B, H, L, rects, am, seq, dead = CASES[name]
    q, k, v = _qkv(len(name), B, H, L)
    table = ops.MaskTable.from_host(rects, np.ones((B, L)) if am is None else am, seq, DEV)
    outs = {}
    for var in (1, 10):
        with _lib.use_lab_attn(var):
            o, lse = ops.mma_attn_core(q, k, v, table, 96 ** -0.5, dead_rows=dead, return_lse=True)
            torch.cuda.synchronize()
            outs[var] = (o.clone(), lse.clone())
    assert torch.equal(outs[1][0], outs[10][0]), f"{name}: {int((outs[1][0] != outs[10][0]).sum())} output elements differ"
    l1, l10 = outs[1][1], outs[10][1]
    assert bool(((l1 == l10) | (torch.isnan(l1) & torch.isnan(l10))).all()), f"{name}: lse differs"


@pytest.mark.parametrize("name", list(CASES))
def test_attn64_shipped_build_vs_exact_f32_kernel(name):
    """Every mask class through the 64-row kernel as shipped (forced at every length), against the exact-f32 kernel."""
    ops = _ops()
    from aki_amd import _lib
    B, H, L, rects, am, seq, dead = CASES[name]
    q, k, v = _qkv(len(name) + 100, B, H, L)
    table = ops.MaskTable.from_host(rects, np.ones((B, L)) if am is None else am, seq, DEV)
    o32 = ops.mma_attn_core(q.float(), k.float(), v.float(), table, 96 ** -0.5, dead_rows=dead)
    with _lib.use_lab_attn(1):
        _, l_ref = ops.mma_attn_core(q, k, v, table, 96 ** -0.5, dead_rows=dead, return_lse=True)     # the 32-row kernel's conventions for dead rows
        torch.cuda.synchronize()
    with _lib.use_lab_attn(9):
        o, lse = ops.mma_attn_core(q, k, v, table, 96 ** -0.5, dead_rows=dead, return_lse=True)
        torch.cuda.synchronize()
    check(n(o), n(o32), torch.bfloat16, f"64-row core, {name}", scale_atol=4.0)
    fin = torch.isfinite(l_ref)
    assert bool((torch.isfinite(lse) == fin).all())
    # m + log l does not depend on which reference maximum m the sums were taken against; the product build's l is the matrix pipe's sum
    # of the bf16-ROUNDED p (the same numbers the P V product uses), the 32-row kernel's the f32 sum of the unrounded ones: 2^-9 per term
    assert float((lse[fin] - l_ref[fin]).abs().max()) < 3e-3


@pytest.mark.parametrize("B,H,L,rects", [(1, 32, 4096, [IMG4]), (2, 8, 2048, [[(6, 150, 150, 2000)]] * 2), (1, 4, 2304, [[(0, 0, 0, 0)]]), (2, 4, 1792, [[(6, 150, 150, 1700)]] * 2)])
def test_attn64_through_the_product_library(B, H, L, rects):
    """What a caller gets: the product library's own choice of core at L >= 1792, against the exact-f32 kernel and (one head) the numpy oracle."""
    ops = _ops()
    q, k, v = _qkv(L, B, H, L)
    table = ops.MaskTable.from_host(rects, np.ones((B, L)), [L] * B, DEV)
    o = ops.mma_attn_core(q, k, v, table, 96 ** -0.5)
    o32 = ops.mma_attn_core(q.float(), k.float(), v.float(), table, 96 ** -0.5)
    check(n(o), n(o32), torch.bfloat16, f"product library B{B} H{H} L{L}", scale_atol=4.0)
    h = H - 1
    sl = lambda a: a[:1, h:h + 1].float().cpu().numpy()
    want = O.mma_attention_core_spans(sl(q), sl(k), sl(v), np.ones((1, L), dtype=np.int64), rects[:1], 96 ** -0.5)
    check(n(o)[:1, :, h * 96:(h + 1) * 96], want, torch.bfloat16, f"product library vs oracle, head {h}", scale_atol=4.0)


@pytest.mark.parametrize("L", [1791, 1792, 1793])
def test_product_rule_boundary_both_cores_agree(L):
    """The product library switches cores at 1792 rows (AKI_ATTN64_MIN_L): one row below, at, and one row above the boundary the output and
    the log-sum-exp must agree with the exact-f32 kernel, and the two cores with each other, on a ragged two-sample batch."""
    ops = _ops()
    from aki_amd import _lib
    B, H = 2, 4
    q, k, v = _qkv(L, B, H, L)
    lens = [L, L - 321]
    table = ops.MaskTable.from_host([[(6, 150, 150, L - 64)], [(6, 150, 150, L - 400)]], _ragged(B, L, lens), lens, DEV)
    o32 = ops.mma_attn_core(q.float(), k.float(), v.float(), table, 96 ** -0.5)
    o, lse = ops.mma_attn_core(q, k, v, table, 96 ** -0.5, return_lse=True)
    check(n(o), n(o32), torch.bfloat16, f"product rule at L = {L}", scale_atol=4.0)
    outs = {}
    for var in (1, 9):
        with _lib.use_lab_attn(var):
            outs[var] = tuple(t.clone() for t in ops.mma_attn_core(q, k, v, table, 96 ** -0.5, return_lse=True))
            torch.cuda.synchronize()
    assert float((outs[1][0].float() - outs[9][0].float()).abs().max()) < 3e-2
    fin = torch.isfinite(outs[1][1])
    assert bool((torch.isfinite(outs[9][1]) == fin).all()) and float((outs[1][1][fin] - outs[9][1][fin]).abs().max()) < 3e-3
    assert torch.equal(o, outs[9][0] if L >= 1792 else outs[1][0]), "the product library did not pick the core its rule names"


def test_attn64_raised_reference_maximum():
    """cdna guide rule 26: the raise of the reference maximum is rare and data dependent, so it gets inputs that force it (a spiked key per
    tile), a full-tensor reference, and three builds that must agree to rounding: the exact one (running maximum per tile), the shipped
    one (blind against the first tile's maximum) and the one that sends every tile through the exact serial path."""
    ops = _ops()
    from aki_amd import _lib
    B, H, L = 1, 4, 2048
    q, k, v = _qkv(26, B, H, L, spike=True)
    table = ops.MaskTable.from_host([[(6, 150, 150, 2000)]], np.ones((B, L)), None, DEV)
    o32 = ops.mma_attn_core(q.float(), k.float(), v.float(), table, 96 ** -0.5)
    outs = {}
    for var in (10, 9, 164):
        with _lib.use_lab_attn(var):
            outs[var] = ops.mma_attn_core(q, k, v, table, 96 ** -0.5).clone()
            torch.cuda.synchronize()
    for var, what in ((10, "exact maximum"), (9, "shipped (blind)"), (164, "exact path on every tile")):
        check(n(outs[var]), n(o32), torch.bfloat16, f"spiked keys, {what}", scale_atol=4.0)
    # the spiked rows themselves (row j + 40 of every tile): their P holds one large value next to tiny ones
    rows = torch.arange(40, L, 64, device=DEV)
    d = (outs[9][:, rows].float() - outs[10][:, rows].float()).abs().max().item()
    assert d < 3e-2, f"spiked rows: shipped vs exact build differ by {d}"


@pytest.mark.parametrize("factor,what", [(8.0, "stays blind: P up to ~2^45 against the first tile's maximum"), (60.0, "row sums overflow: verification fails, rank walked again")])
def test_attn64_blind_softmax_verification(factor, what):
    """Keys from row 1024 on are scaled up: every later tile's scores exceed the reference maximum the rank took from its first tile.
    Below the bound (row sum < 2^64) the blind pass is exact to rounding - bf16 has f32's exponent range; above it exp2 overflows,
    the row sum is inf, and the kernel must notice at the rank's end and walk the rank again through the exact path (without that
    the outputs are NaN).  Both against the exact-f32 kernel, through the product library's own choice of core."""
    ops = _ops()
    B, H, L = 2, 4, 2304
    q, k, v = _qkv(int(factor), B, H, L)
    k[:, :, 1024:] *= factor
    table = ops.MaskTable.from_host([[(6, 150, 150, 2200)]] * B, np.ones((B, L)), None, DEV)
    o, lse = ops.mma_attn_core(q, k, v, table, 96 ** -0.5, return_lse=True)
    o32, lse32 = ops.mma_attn_core(q.float(), k.float(), v.float(), table, 96 ** -0.5, return_lse=True)
    assert torch.isfinite(o.float()).all(), what
    check(n(o), n(o32), torch.bfloat16, f"keys x{factor}: {what}", scale_atol=4.0)
    # log-sum-exp in natural units: scores reach several hundred here, compare relatively
    assert float(((lse - lse32).abs() / (1.0 + lse32.abs())).max()) < 2e-2


def test_attn64_is_run_to_run_deterministic():
    """Race / hazard screen at the long-context shape: six launches must agree bit for bit (outputs and lse).  The kernel's MFMAs are
    inline asm: hipcc pads no hazard around them (tools/attn64_hazards.py scans the code object; this is the hardware-side check)."""
    ops = _ops()
    B, H, L = 2, 32, 4096
    q, k, v = _qkv(64, B, H, L)
    table = ops.MaskTable.from_host([IMG4] * B, _ragged(B, L, [L, L - 300]), [L, L - 300], DEV)
    o0, l0 = ops.mma_attn_core(q, k, v, table, 96 ** -0.5, return_lse=True)
    o0, l0 = o0.clone(), l0.clone()
    for i in range(5):
        o, lse = ops.mma_attn_core(q, k, v, table, 96 ** -0.5, return_lse=True)
        assert torch.equal(o, o0), f"launch {i + 1}: {int((o != o0).sum())} output elements differ from launch 0"
        assert bool(((lse == l0) | (torch.isnan(lse) & torch.isnan(l0))).all()), f"launch {i + 1}: lse differs"


def test_attn64_kv_cache_capacity():
    """K / V handed over as a KV cache whose capacity exceeds L (prefill writes the first L rows): rows past L must never be read as data
    (the 64-row kernel's buffer descriptor ends at row L; the cache's tail is filled with NaN here)."""
    ops = _ops()
    B, H, L, cap = 1, 4, 2100, 2304
    q, k, v = _qkv(7, B, H, L)
    kc = torch.full((B, H, cap, 96), float("nan"), device=DEV, dtype=torch.bfloat16)
    vc = torch.full((B, H, cap, 96), float("nan"), device=DEV, dtype=torch.bfloat16)
    kc[:, :, :L] = k
    vc[:, :, :L] = v
    table = ops.MaskTable.from_host([[(6, 150, 150, 2000)]], np.ones((B, L)), None, DEV)
    o = ops.mma_attn_core(q, kc, vc, table, 96 ** -0.5)
    o_ref = ops.mma_attn_core(q, k, v, table, 96 ** -0.5)
    assert torch.isfinite(o.float()).all()
    assert torch.equal(o, o_ref)
