"""The layer loops issued from one C call (csrc/stack.hip: aki_decoder_stack_fwd, aki_siglip_stack_fwd) against the per-layer Python loops
they replace (phi3.py / siglip.py forward_folded): the same launches with the same arguments, so every output is compared BIT FOR BIT -
hidden states, the folded final statistics' effect on the logits, and the KV cache a prefill leaves behind."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _lm(n_layers, seed, **kw):
    from aki_amd.phi3 import Phi3ForCausalLM, make_phi3_config
    torch.manual_seed(seed)
    cfg = make_phi3_config(num_hidden_layers=n_layers, vocab_size=2048, pad_token_id=0, eos_token_id=2, **kw)
    lm = Phi3ForCausalLM(cfg)
    g = torch.Generator().manual_seed(seed)
    for n, p in lm.named_parameters():
        p.data.copy_(1.0 + 0.2 * torch.randn(p.shape, generator=g) if p.dim() == 1 else torch.randn(p.shape, generator=g) * 0.02)
    return lm.to(DEV).to(torch.bfloat16).eval(), cfg


@pytest.mark.parametrize("B,L,width", [(1, 655, "full"), (1, 207, "full"), (2, 150, "small"), (3, 64, "small")])
@pytest.mark.parametrize("use_cache", [False, True])
def test_decoder_stack_equals_the_per_layer_loop(B, L, width, use_cache):
    from aki_amd import ops
    kw = {} if width == "full" else dict(hidden_size=384, intermediate_size=1024, num_attention_heads=4, num_key_value_heads=4)
    lm, cfg = _lm(3, seed=11, **kw)
    g = torch.Generator().manual_seed(L)
    x = (torch.randn(B, L, cfg.hidden_size, generator=g) * 0.5).to(torch.bfloat16).to(DEV)
    lens = [L - 7 * b for b in range(B)]
    mask = np.zeros((B, L), dtype=bool)
    for b in range(B):
        mask[b, : lens[b]] = True
    rects = [[(3, min(40, lens[b] // 2), min(40, lens[b] // 2), lens[b] - 5)] for b in range(B)]
    table = ops.MaskTable.from_host(rects, mask, lens, DEV)
    out = {}
    for stack in (True, False):
        lm.model.use_layer_stack = stack
        with torch.no_grad():
            o = lm(inputs_embeds=x, attention_mask=table, use_cache=use_cache, cache_capacity=L + 9)
        out[stack] = (o.logits.clone(), o.past_key_values)
    lm.model.use_layer_stack = True
    assert torch.equal(out[True][0], out[False][0]), "logits differ between the one-call layer loop and the per-layer launches"
    if use_cache:
        ca, cb = out[True][1], out[False][1]
        assert torch.equal(ca.cache_len, cb.cache_len)
        for i in range(len(ca.k)):
            assert torch.equal(ca.k[i][:, :, :L], cb.k[i][:, :, :L]) and torch.equal(ca.v[i][:, :, :L], cb.v[i][:, :, :L]), f"KV cache of layer {i}"


def test_decoder_stack_is_skipped_when_python_must_run_between_layers():
    """Module hooks on a layer (the sharded trainer gathers weights there) and an installed event tap switch the one-call loop off."""
    from aki_amd import ops
    lm, cfg = _lm(2, seed=5, hidden_size=384, intermediate_size=1024, num_attention_heads=4, num_key_value_heads=4)
    x = torch.zeros(1, 64, 384, dtype=torch.bfloat16, device=DEV)
    assert lm.model._can_stack(x)
    h = lm.model.layers[1].register_forward_pre_hook(lambda m, a: None)
    assert not lm.model._can_stack(x)
    h.remove()
    ops.set_event_tap(ops.EventTap(tags=("linear",)))
    try:
        assert not lm.model._can_stack(x)
    finally:
        ops.set_event_tap(None)
    assert lm.model._can_stack(x)


@pytest.mark.parametrize("N,size,width", [(1, 336, "full"), (2, 224, "small"), (3, 56, "small")])
def test_siglip_stack_equals_the_per_layer_loop(N, size, width):
    from aki_amd.siglip import SiglipVisionTransformer, make_siglip_config
    kw = dict(num_hidden_layers=3, image_size=size)
    if width == "small":
        kw.update(hidden_size=576, intermediate_size=1000, num_attention_heads=8)
    torch.manual_seed(3)
    vt = SiglipVisionTransformer(make_siglip_config(**kw))
    g = torch.Generator().manual_seed(7)
    for n, p in vt.named_parameters():
        if "layer_norm" in n or "layernorm" in n:
            p.data.copy_((1.0 if n.endswith("weight") else 0.0) + 0.2 * torch.randn(p.shape, generator=g))
        elif p.dim() == 1:
            p.data.copy_(0.1 * torch.randn(p.shape, generator=g))
    vt = vt.to(DEV).to(torch.bfloat16).eval()
    px = ((torch.rand((N, 3, size, size), generator=g) - 0.5) / 0.5).to(DEV, torch.bfloat16)
    out = {}
    for stack in (True, False):
        vt.encoder.use_layer_stack = stack
        with torch.no_grad():
            out[stack] = vt(px).last_hidden_state.clone()
    vt.encoder.use_layer_stack = True
    assert torch.isfinite(out[True].float()).all()
    assert torch.equal(out[True], out[False]), "SigLIP tower differs between the one-call layer loop and the per-layer launches"


def test_stack_tables_follow_weight_changes():
    """The one-call loops prepare their per-layer tables (gain-folded weights, pointers) once and re-check a cheap signature per call.  Every
    way a weight can change must invalidate it: an in-place torch write, a re-allocation (`.data = ...`), a trainer-style raw write announced
    by the weight epoch, and model.train() must drop the folded copies."""
    from aki_amd import train_ops as T
    lm, cfg = _lm(2, seed=3, hidden_size=384, intermediate_size=1024, num_attention_heads=4, num_key_value_heads=4)
    x = (torch.randn(1, 80, 384, generator=torch.Generator().manual_seed(1)) * 0.5).to(torch.bfloat16).to(DEV)

    def both():
        out = []
        for stack in (True, False):
            lm.model.use_layer_stack = stack
            with torch.no_grad():
                out.append(lm(inputs_embeds=x).logits.clone())
        lm.model.use_layer_stack = True
        return out

    a, b = both()
    assert torch.equal(a, b)
    w = lm.model.layers[1].mlp.down_proj.weight
    with torch.no_grad():
        w.mul_(1.5)                                          # in place: version counter
    a2, b2 = both()
    assert torch.equal(a2, b2) and not torch.equal(a2, a)
    g = lm.model.layers[0].input_layernorm.weight
    g.data = (g.data * 0.5).clone()                          # re-allocated: another address (the gain is folded into the cached qkv weight)
    a3, b3 = both()
    assert torch.equal(a3, b3) and not torch.equal(a3, a2)
    lm.model.layers[1].self_attn.o_proj.weight.data.view(-1)[:1000].zero_()       # a write torch does not count on the Parameter ...
    T.bump_weight_epoch()                                    # ... announced the way the trainers announce theirs
    a4, b4 = both()
    assert torch.equal(a4, b4)
    assert lm.model._stack_table is not None and lm.model._stack_table.keep is not None
    lm.train()
    assert lm.model._stack_table is None
    lm.eval()


def test_siglip_and_perceiver_stacks_follow_replaced_parameter_objects():
    """load_state_dict(assign=True) and `module.weight = nn.Parameter(...)` REPLACE parameter objects; the old objects
    stay alive with an unchanged (address, version), so a signature over a cached object list keeps the one-call loops on the old weights.
    After each kind of replacement the stacked forward must equal the per-layer path on the new weights - and differ from before."""
    import copy
    from aki_amd.siglip import SiglipVisionTransformer, make_siglip_config
    from aki_amd.helpers import PerceiverResampler
    torch.manual_seed(11)
    vt = SiglipVisionTransformer(make_siglip_config(num_hidden_layers=2, image_size=56, hidden_size=576, intermediate_size=1000, num_attention_heads=8))
    vt = vt.to(DEV).to(torch.bfloat16).eval()
    px = ((torch.rand((2, 3, 56, 56)) - 0.5) / 0.5).to(DEV, torch.bfloat16)

    def tower(stack):
        vt.encoder.use_layer_stack = stack
        with torch.no_grad():
            return vt(px).last_hidden_state.clone()

    a = tower(True)
    sd = {k: (v.clone() * 1.25 if v.dim() == 2 else v.clone()) for k, v in vt.state_dict().items()}
    vt.load_state_dict(sd, assign=True)                                     # new Parameter objects around new storage
    b_stack, b_loop = tower(True), tower(False)
    assert torch.equal(b_stack, b_loop), "SigLIP stack kept the weights from before load_state_dict(assign=True)"
    assert not torch.equal(b_stack, a)
    fc1 = vt.encoder.layers[1].mlp.fc1
    fc1.weight = torch.nn.Parameter(fc1.weight.detach() * 0.5, requires_grad=False)      # attribute assignment on a sub-module
    c_stack, c_loop = tower(True), tower(False)
    assert torch.equal(c_stack, c_loop) and not torch.equal(c_stack, b_stack)
    vt.encoder.use_layer_stack = True

    pr = PerceiverResampler(dim=576, dim_inner=384, depth=2, dim_head=64, heads=8, num_latents=16, ff_mult=2).to(DEV).to(torch.bfloat16).eval()
    x = (torch.randn(1, 1, 1, 40, 576) * 0.5).to(DEV, torch.bfloat16)

    def conn(stack):
        pr.use_layer_stack = stack
        with torch.no_grad():
            return pr(x).clone()

    a = conn(True)
    sd = {k: (v.clone() * 1.5 if v.dim() == 2 else v.clone()) for k, v in pr.state_dict().items()}
    pr.load_state_dict(sd, assign=True)
    b_stack, b_loop = conn(True), conn(False)
    assert torch.equal(b_stack, b_loop), "Perceiver stack kept the weights from before load_state_dict(assign=True)"
    assert not torch.equal(b_stack, a)
    pr.use_layer_stack = True


@pytest.mark.parametrize("n1,proj", [(576, True), (729, True), (100, False)])
def test_perceiver_stack_equals_the_python_loop(n1, proj):
    """aki_perceiver_stack_fwd (one (sample, image) pair) against PerceiverResampler's own inference loop: bit for bit, with and without the final
    projection, LayerNorm gains / biases randomised."""
    from aki_amd.helpers import PerceiverResampler
    torch.manual_seed(5)
    pr = PerceiverResampler(dim=1152, dim_inner=3072 if proj else None, num_latents=144)
    g = torch.Generator().manual_seed(8)
    for n_, p in pr.named_parameters():
        if p.dim() == 1:
            p.data.copy_((1.0 if n_.endswith("weight") else 0.0) + 0.2 * torch.randn(p.shape, generator=g))
        elif n_ != "latents":
            p.data.copy_(torch.randn(p.shape, generator=g) * 0.03)
    pr = pr.to(DEV).to(torch.bfloat16).eval()
    x = (torch.randn(1, 1, 1, n1, 1152, generator=g)).to(torch.bfloat16).to(DEV)
    out = {}
    for stack in (True, False):
        pr.use_layer_stack = stack
        with torch.no_grad():
            out[stack] = pr(x).clone()
    pr.use_layer_stack = True
    assert out[True].shape == (1, 1, 144, 3072 if proj else 1152) and torch.isfinite(out[True].float()).all()
    assert torch.equal(out[True], out[False])
    with torch.no_grad():                                   # two pairs: the Python loop (the stack serves one pair)
        two = pr(torch.cat([x, x], 0))
    assert torch.equal(two[0], out[True][0]) and torch.equal(two[1], out[True][0])
