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
