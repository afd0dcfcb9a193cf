"""First-line check and A/B of the 64-row attention core (mma_attn64_bf16.hip) against the 32-row kernel, in one process on the lab library:
   variant 1  = 32-row kernel (two waves per SIMD)            - the reference for bit-identity
   variant 10 = 64-row kernel, exact running maximum (THR 0)  - must equal variant 1 bit for bit (outputs and lse)
   variant 9  = 64-row kernel as shipped (THR 8)              - compared with the exact-f32 kernel under the suite's bf16 bar
Mask cases: no image, one image, four images, ragged lengths with stacking padding (dead rows, both conventions), left padding with
holes, lengths that are no multiple of 32 / 64 / 256, more than 128 blocks (position order).  Then interleaved timing.
    python tools/attn64_check.py [--time-only] [--quick]"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from aki_amd import ops, _lib

lab = _lib.load_lab()
_lib._lib = lab
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
TIME_ONLY = "--time-only" in sys.argv
QUICK = "--quick" in sys.argv


def pairs_of(L, rects):
    p = L * (L + 1) // 2
    for (r0, r1, c0, c1) in rects:
        for r in range(r0, r1):
            p += max(0, c1 - max(c0, r + 1))
    return p


def run(q, k, v, table, n, dead=None):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        o = ops.mma_attn_core(q, k, v, table, 96 ** -0.5)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3, o


def case(name, B, H, L, rects, am=None, seq=None, dead_rows=None, spike=False):
    q, k, v = (torch.randn(B, H, L, 96, device=dev, generator=g).to(torch.bfloat16) for _ in range(3))
    if spike:   # one key per 64 that every later tile beats by a wide margin: forces the reference maximum to be raised many times
        for j in range(0, L, 64):
            k[:, :, j] = q[:, :, min(L - 1, j + 40)] * (0.5 + 0.4 * j / L)
    am = np.ones((B, L)) if am is None else am
    table = ops.MaskTable.from_host(rects, am, seq, dev)
    kw = {} if dead_rows is None else {"dead_rows": dead_rows}
    outs = {}
    for var in (1, 10, 9):
        lab.aki_lab_set_attn_variant(var)
        o, lse = ops.mma_attn_core(q, k, v, table, 96 ** -0.5, return_lse=True, **kw)
        torch.cuda.synchronize()
        outs[var] = (o.clone(), lse.clone())
    lab.aki_lab_set_attn_variant(1)
    o32 = ops.mma_attn_core(q.float(), k.float(), v.float(), table, 96 ** -0.5, **kw)
    lab.aki_lab_set_attn_variant(0)
    same_o = torch.equal(outs[1][0], outs[10][0])
    l1, l10 = outs[1][1], outs[10][1]
    same_l = bool(((l1 == l10) | (torch.isnan(l1) & torch.isnan(l10))).all())
    nbad = int((outs[1][0] != outs[10][0]).sum())
    e1 = (outs[1][0].float() - o32).abs().max().item()
    e9 = (outs[9][0].float() - o32).abs().max().item()
    m1 = (outs[1][0].float() - o32).abs().mean().item()
    m9 = (outs[9][0].float() - o32).abs().mean().item()
    lse9 = (outs[9][1] - l1)
    lse9 = lse9[torch.isfinite(lse9)].abs().max().item() if torch.isfinite(lse9).any() else 0.0
    nan9 = int(torch.isnan(outs[9][0].float()).sum())
    ok = same_o and same_l and nan9 == 0 and e9 <= max(2.0 * e1, 4e-2)
    print(f"{'ok  ' if ok else 'FAIL'} {name:34s} B{B} H{H} L{L}: exact == 32-row: O {same_o} ({nbad} differ) lse {same_l} | vs f32 kernel max/mean: 32-row {e1:.4f}/{m1:.5f}  "
          f"64-row {e9:.4f}/{m9:.5f}  lse diff {lse9:.2e} nan {nan9}", flush=True)
    return ok


def timing():
    CASES = [(1, 32, 4096, [[(6, 150, 150, 4032), (900, 1044, 1044, 4032), (1800, 1944, 1944, 4032), (2700, 2844, 2844, 4032)]]),
             (4, 32, 4096, [[(6, 150, 150, 4032), (900, 1044, 1044, 4032), (1800, 1944, 1944, 4032), (2700, 2844, 2844, 4032)]] * 4),
             (1, 32, 4096, [[(0, 0, 0, 0)]]), (4, 32, 4096, [[(0, 0, 0, 0)]] * 4),
             (1, 32, 2048, [[(6, 150, 150, 2000)]]), (4, 32, 2048, [[(6, 150, 150, 2000)]] * 4),
             (8, 32, 1024, [[(6, 150, 150, 1000)]] * 8), (8, 32, 655, [[(6, 150, 150, 638)]] * 8), (1, 32, 655, [[(6, 150, 150, 638)]])]
    res = []
    for (B, H, L, rects) in CASES:
        q, k, v = (torch.randn(B, H, L, 96, device=dev, generator=g).to(torch.bfloat16) for _ in range(3))
        table = ops.MaskTable.from_host(rects, np.ones((B, L)), None, dev)
        best = {1: 1e9, 9: 1e9, 10: 1e9}
        for r in range(5):
            for var in (1, 9, 10):
                lab.aki_lab_set_attn_variant(var)
                t, _ = run(q, k, v, table, 10)
                best[var] = min(best[var], t)
        lab.aki_lab_set_attn_variant(0)
        fl = 4.0 * 96 * pairs_of(L, rects[0]) * B * H
        tf = {v_: fl / best[v_] / 1e6 for v_ in best}
        # product rule (variant 0) and the vendor yardstick: torch SDPA, causal only (it has no span mask: fewer visible pairs than the MMA mask)
        lab.aki_lab_set_attn_variant(0)
        t_prod = min(run(q, k, v, table, 10)[0] for _ in range(3))
        import torch.nn.functional as F
        def sdpa(n):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(n):
                F.scaled_dot_product_attention(q, k, v, is_causal=True)
            b.record()
            torch.cuda.synchronize()
            return a.elapsed_time(b) / n * 1e3
        try:
            sdpa(3)
            t_sdpa = min(sdpa(10) for _ in range(3))
            fl_c = 4.0 * 96 * (L * (L + 1) // 2) * B * H
            sd = f" | torch SDPA causal {t_sdpa:7.1f} us {fl_c / t_sdpa / 1e6:5.0f} TF/s"
        except Exception as e:      # noqa: BLE001
            t_sdpa, sd = None, f" | torch SDPA: {type(e).__name__}"
        print(f"B{B} H{H} L{L}: 32-row {best[1]:7.1f} us {tf[1]:5.0f} TF/s ({tf[1]/2500:.3f}) | 64-row {best[9]:7.1f} us {tf[9]:5.0f} TF/s ({tf[9]/2500:.3f}) | 64-row exact {best[10]:7.1f} us "
              f"{tf[10]:5.0f} TF/s | 64/32 time {best[9]/best[1]:.3f} | product rule {t_prod:7.1f} us ({fl/t_prod/1e6/2500:.3f})" + sd, flush=True)
        res.append({"B": B, "L": L, "us_32row": best[1], "us_64row": best[9], "us_64row_exact": best[10], "tf_32row": tf[1], "tf_64row": tf[9], "us_product": t_prod, "us_torch_sdpa_causal": t_sdpa})
    return res


ok = True
if not TIME_ONLY:
    IMG4 = [(6, 150, 150, 4032), (900, 1044, 1044, 4032), (1800, 1944, 1944, 4032), (2700, 2844, 2844, 4032)]
    ok &= case("causal only", 1, 2, 512, [[(0, 0, 0, 0)]])
    ok &= case("one image", 2, 2, 655, [[(6, 150, 150, 638)]] * 2)
    ok &= case("short, odd length", 2, 3, 207, [[(6, 150, 150, 190)]] * 2)
    ok &= case("tiny", 1, 1, 32, [[(0, 0, 0, 0)]])
    ok &= case("tiny 40", 2, 2, 40, [[(3, 19, 19, 33)]] * 2)
    ok &= case("L=1000", 1, 2, 1000, [[(10, 154, 154, 980)]])
    ok &= case("four images L=4096", 1, 4, 4096, [IMG4])
    ok &= case("four images L=4096 B=2 H=32", 2, 32, 4096, [IMG4] * 2)
    ok &= case("spiked keys (maximum raised often)", 1, 4, 2048, [[(6, 150, 150, 2000)]], spike=True)
    ok &= case("L=4100 (129 blocks: position order)", 1, 2, 4100, [IMG4])
    ok &= case("L=5000 position order", 1, 2, 5000, [[(6, 150, 150, 4900), (3000, 3144, 3144, 4900)]])
    # ragged lengths: stacking padding (rows >= seq_len dead) under both conventions
    B, L = 3, 1500
    am = np.ones((B, L)); seq = [L, 1111, 700]
    for b in range(B):
        am[b, seq[b]:] = 0
    rects = [[(6, 150, 150, min(1400, seq[b] - 17))] for b in range(B)]
    for dr in (1, 0):
        ok &= case(f"ragged + dead rows ({dr})", B, 2, L, rects, am, seq, dead_rows=dr)
    # left padding with a hole
    B, L = 2, 1200
    am = np.ones((B, L)); am[0, :137] = 0; am[1, :64] = 0; am[1, 500:520] = 0
    ok &= case("left padding + hole", B, 2, L, [[(150, 294, 294, 1100)], [(70, 214, 214, 1150)]], am, None)
    print("ALL OK" if ok else "SOME FAILED", flush=True)
if not QUICK:
    res = timing()
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump({"ok": bool(ok), "timing": res}, open("gpurun_out/attn64_check.json", "w"), indent=1)
