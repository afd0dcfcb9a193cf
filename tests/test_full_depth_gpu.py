"""BASELINE configs[1] at FULL DEPTH: AKI-4B with all 32 Phi-3.5-mini decoder layers and all 27 SigLIP layers (the thing
bench.py runs), bf16 HIP logits against the fp32 torch oracle (oracle/aki_torch.py, pinned to the reference) evaluated on
the host with the SAME bf16-rounded weights and inputs.  What the shallower tests cannot see is how the bf16 error grows
over 32 + 27 layers; the yardstick is the oracle itself evaluated in bf16 (= the reference's `model.to(bfloat16)` eager
path, train/train.py:270-271) against its fp32 self.  ~17 GB of host RAM for the fp32 weights, about a minute of CPU."""
import json
import os
import time

import numpy as np
import pytest
import torch

from conftest import ROOT, record_parity

pytestmark = pytest.mark.gpu
DEV = "cuda"
N_TXT, NV, PX = 512, 144, 336


def _prompts(B, media_id, seed):
    """SURVEY 8(d) synthetic chat prompts (as bench.py); sample 1 is the padded variant (0.8 * N_txt real tokens)."""
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(3, 32000, (B, N_TXT), generator=g)
    am = torch.ones_like(ids)
    for b in range(B):
        n = N_TXT if b != 1 else int(0.8 * N_TXT)
        ids[b, 0], ids[b, 6], ids[b, n - 18], ids[b, n - 17], ids[b, n - 1] = 1, media_id, 32007, 32001, 2
        ids[b, n:] = 32000
        am[b, n:] = 0
    vx = (torch.rand((B, 1, 1, 3, PX, PX), generator=g) - 0.5) / 0.5
    return vx, ids, am


@pytest.mark.timeout(3000)
def test_aki4b_full_depth_bf16_logits_vs_fp32_oracle():
    import aki_torch as OT
    from aki_amd.factory import build_aki
    B = 2
    m = build_aki(dtype=torch.bfloat16, device=DEV, seed=7).eval()           # AKI-4B: 32 + 27 layers, real widths
    assert len(m.lang_model.model.layers) == 32 and len(m.vision_encoder.encoder.layers) == 27
    vx, ids, am = _prompts(B, m.media_token_id, 11)
    vx16 = vx.to(torch.bfloat16)
    with torch.no_grad():
        got = m(vx16.to(DEV), ids.to(DEV), attention_mask=am.to(DEV)).logits.float().cpu()
    L = N_TXT - 1 + NV
    assert got.shape == (B, L, 32011 + 2) and bool(torch.isfinite(got).all())
    p16 = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    del m
    torch.cuda.empty_cache()
    cfg = dict(vis_layers=27, vis_heads=16, lm_layers=32, lm_heads=32, max_original_id=32010, media_token_id=32011,
               pad_token_id=32000, num_vision_tokens=NV)
    torch.set_num_threads(min(32, os.cpu_count() or 8))
    t0 = time.time()
    with torch.no_grad():
        p32 = {k: v.float() for k, v in p16.items()}
        want = OT.aki_forward(p32, cfg, vx16.float(), ids, am)
        ref = want["logits"]
        del p32
        t32 = time.time() - t0
        t0 = time.time()
        # yardstick: the same eager graph in bf16 (what `--precision bf16` makes the reference compute)
        ref16 = OT.aki_forward(p16, cfg, vx16, ids, am)["logits"].float()
        t16 = time.time() - t0
    # rows of real tokens (the splice keeps padded positions as rows; they are compared too, separately)
    valid = torch.from_numpy(np.asarray(want["prep"]["mask_1d"]).astype(bool))            # [B, L]
    mx = max(1.0, float(ref.abs().max()))
    e_hip, e_ref = (got - ref).abs(), (ref16 - ref).abs()
    stats = {}
    for name, rows in (("valid rows", valid), ("padded rows", ~valid)):
        if not bool(rows.any()):
            continue
        stats[name] = dict(hip_max=float(e_hip[rows].max()), hip_mean=float(e_hip[rows].mean()),
                           eager_bf16_max=float(e_ref[rows].max()), eager_bf16_mean=float(e_ref[rows].mean()))
    top_ref = ref.argmax(-1)
    stats["argmax_agreement_hip"] = float((got.argmax(-1) == top_ref)[valid].float().mean())
    stats["argmax_agreement_eager_bf16"] = float((ref16.argmax(-1) == top_ref)[valid].float().mean())
    stats.update(max_abs_ref=mx, seconds_fp32_oracle=round(t32, 1), seconds_bf16_oracle=round(t16, 1), batch=B, L=L)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "parity_full_depth.json"), "w") as f:
        json.dump(stats, f, indent=1)
    print(json.dumps(stats))
    v = stats["valid rows"]
    record_parity("AKI-4B 32+27 layers logits, valid rows", torch.bfloat16, v["hip_max"], v["hip_mean"], mx,
                  "<= 1.5x mean / 2x max of the oracle's own bf16-eager error + 1e-3*max|ref|")
    assert v["hip_mean"] <= 1.5 * v["eager_bf16_mean"] + 1e-3 * mx, stats
    assert v["hip_max"] <= 2.0 * v["eager_bf16_max"] + 1e-2 * mx, stats
    if "padded rows" in stats:                     # rows of padding tokens: finite, same convention (uniform softmax rows exist only beyond seq_len)
        pr = stats["padded rows"]
        assert pr["hip_mean"] <= 1.5 * pr["eager_bf16_mean"] + 1e-3 * mx, stats


@pytest.mark.timeout(3000)
def test_aki4b_full_depth_fp8_logits_vs_fp32_oracle_and_bf16():
    """BASELINE configs[4] at the workload's width and depth: AKI-4B, 32 decoder layers with `enable_fp8()`
    (e4m3 projections, per-token / per-weight-row scales; attention, residual stream and the vision side stay bf16), batch 2,
    336 px + 512 tokens.  No reference behaviour exists for fp8 (parity unpinned): the yardsticks are the fp32 oracle on the same
    bf16-rounded weights and this library's own bf16 path.  Three configurations: every projection + the head in e4m3 (the
    benchmark's `--dtype fp8`), head in bf16, head + the residual-stream writers (o_proj, down_proj) in bf16.
    What the numbers say (gpurun_out/parity_full_depth_fp8.json -> profiles/): an e4m3 GEMM output carries a few percent of
    rounding noise, 128 such outputs feed the residual stream and the head puts its own straight on the logits; with RANDOM-INIT
    weights the logits are nearly flat (top-1 / top-2 margins of a few 1e-2), so the arg-max agreement with fp32 collapses
    (0.39 against bf16's 0.93) although the relative L2 error is the expected ~0.3.  Asserted: finite logits, the error inside the
    random-walk model of the e4m3 noise, and that it shrinks as projections are moved back to bf16 - the arg-max figures are
    recorded, not asserted (a trained checkpoint, which this build cannot load offline, is what they would be meaningful on)."""
    import aki_torch as OT
    from aki_amd.factory import build_aki
    B = 2
    m = build_aki(dtype=torch.bfloat16, device=DEV, seed=7).eval()
    vx, ids, am = _prompts(B, m.media_token_id, 11)
    vx16 = vx.to(torch.bfloat16)
    run = lambda: m(vx16.to(DEV), ids.to(DEV), attention_mask=am.to(DEV)).logits.float().cpu()
    got = {}
    with torch.no_grad():
        got["bf16"] = run()
        m.lang_model.enable_fp8()
        got["fp8"] = run()
        m.lang_model.enable_fp8(True, head=False)
        got["fp8_head_bf16"] = run()
        m.lang_model.enable_fp8(True, head=False, residual_writers=False)
        got["fp8_qkv_gateup_only"] = run()
        m.lang_model.enable_fp8(False)
    assert all(bool(torch.isfinite(g).all()) for g in got.values())
    p16 = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    del m
    torch.cuda.empty_cache()
    cfg = dict(vis_layers=27, vis_heads=16, lm_layers=32, lm_heads=32, max_original_id=32010, media_token_id=32011,
               pad_token_id=32000, num_vision_tokens=NV)
    torch.set_num_threads(min(32, os.cpu_count() or 8))
    with torch.no_grad():
        want = OT.aki_forward({k: v.float() for k, v in p16.items()}, cfg, vx16.float(), ids, am)
    ref = want["logits"]
    valid = torch.from_numpy(np.asarray(want["prep"]["mask_1d"]).astype(bool))
    mx = max(1.0, float(ref.abs().max()))
    top = ref.argmax(-1)
    top2 = ref.topk(2, dim=-1).values
    stats = dict(batch=B, L=N_TXT - 1 + NV, max_abs_ref=mx, ref_logit_std=float(ref[valid].std()),
                 ref_median_top1_top2_margin=float((top2[..., 0] - top2[..., 1])[valid].median()))
    for k, g in got.items():
        e = (g - ref).abs()[valid]
        stats[k] = dict(max=float(e.max()), mean=float(e.mean()), rel_l2=float((g - ref)[valid].norm() / ref[valid].norm()),
                        argmax_agreement=float((g.argmax(-1) == top)[valid].float().mean()))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "parity_full_depth_fp8.json"), "w") as f:
        json.dump(stats, f, indent=1)
    print(json.dumps(stats))
    record_parity("AKI-4B 32+27 layers logits, fp8 projections + head, valid rows", torch.bfloat16, stats["fp8"]["max"], stats["fp8"]["mean"], mx,
                  "rel L2 <= 0.5 (random walk of e4m3 noise over 129 GEMMs); no reference for fp8")
    assert stats["fp8"]["rel_l2"] <= 0.5, stats
    assert stats["fp8_qkv_gateup_only"]["rel_l2"] < stats["fp8_head_bf16"]["rel_l2"] <= stats["fp8"]["rel_l2"] * 1.02, stats
    assert stats["fp8_qkv_gateup_only"]["argmax_agreement"] >= stats["fp8"]["argmax_agreement"], stats


@pytest.mark.timeout(3000)
def test_aki4b_full_size_training_step_properties():
    """BASELINE configs[2]'s per-GPU workload in the suite the driver runs: AKI-4B, all 32 + 27 layers, batch 8 x
    (336 px image + 512-token prompt), forward + backward + clip 1.0 + AdamW on the HIP kernels.  There is no oracle at this size
    that finishes in seconds, so the checks are properties: the loss is finite and near ln(vocab) for random-init weights, it
    DEcreases over three optimizer steps on the same batch, the global gradient norm is finite and positive, and a second run
    from the same initial weights reproduces losses and gradients bit for bit (every kernel on the path is deterministic)."""
    import bench
    from aki_amd.factory import build_aki
    from aki_amd.trainer import AkiTrainer

    def run():
        m = build_aki(dtype=torch.bfloat16, device=DEV, seed=3)
        m.train()
        m.set_trainable()
        tr = AkiTrainer(m, lr=2e-4, betas=(0.9, 0.999), weight_decay=0.01, max_grad_norm=1.0)
        vx, ids, am = bench.synth_batch(8, torch.device(DEV), torch.bfloat16, m.media_token_id, seed=5)
        labels = ids.clone()
        labels[labels == m.media_token_id] = -100
        losses, norms = [], []
        g_first = None
        for i in range(3):
            tr.zero_grad()
            out = m(vx, ids, attention_mask=am, labels=labels)
            tr.backward(out.loss)
            if i == 0:
                g_first = tr.g16.clone()
            tr.optimizer_step()
            losses.append(float(out.loss.detach()))
            norms.append(float(tr.grad_norm()))
        w_end = tr.w16.clone()
        del tr, m
        torch.cuda.empty_cache()
        return losses, norms, g_first, w_end

    l1, n1, g1, w1 = run()
    assert all(np.isfinite(l1)) and all(np.isfinite(n1)) and min(n1) > 0.0, (l1, n1)
    assert abs(l1[0] - np.log(32013.0)) < 1.0, l1                 # random init: about ln(vocab)
    assert l1[2] < l1[1] < l1[0], f"loss does not decrease over three steps: {l1}"
    l2, n2, g2, w2 = run()
    assert l1 == l2 and n1 == n2, (l1, l2, n1, n2)
    assert torch.equal(g1, g2), f"{int((g1 != g2).sum())} gradient elements differ between two runs"
    assert torch.equal(w1, w2), "weights after three steps differ between two runs"
