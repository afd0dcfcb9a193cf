"""CPU checks of the host-side algebra behind the folded normalisation (aki_amd/ops.py::fold_gain, aki_amd/siglip.py::fold_layernorm,
aki_amd/ops.py::Prepared): the identities the GPU epilogues rely on, in float64, and the cache's invalidation rules.
No kernel is called - the C ABI side is covered by tests/test_norm_fold_gpu.py on the GPU box."""
import torch

from aki_amd import ops
from aki_amd.siglip import fold_layernorm


def test_rmsnorm_commutes_with_the_projection():
    """RMSNorm(x; gamma) @ W^T == rstd[m] * (x @ (W diag(gamma))^T)   (HF:phi3/modeling_phi3.py:266-284 followed by nn.Linear)."""
    g = torch.Generator().manual_seed(0)
    x = torch.randn(7, 96, generator=g, dtype=torch.float64) * 3
    w = torch.randn(40, 96, generator=g, dtype=torch.float64)
    gamma = 1 + 0.3 * torch.randn(96, generator=g, dtype=torch.float64)
    eps = 1e-5
    rstd = torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + eps)
    want = (x * rstd * gamma) @ w.t()
    got = rstd * (x @ ops.fold_gain(w, gamma).t())
    assert torch.allclose(got, want, rtol=1e-12, atol=1e-12)


def test_layernorm_commutes_with_the_projection():
    """LayerNorm(x; gamma, beta) @ W^T + b == rstd * (x @ W'^T - mean * c) + (W beta + b)   (HF:siglip/modeling_siglip.py:329-354)."""
    g = torch.Generator().manual_seed(1)
    x = torch.randn(9, 64, generator=g, dtype=torch.float64) * 2 + 5
    w = torch.randn(24, 64, generator=g, dtype=torch.float64)
    b = torch.randn(24, generator=g, dtype=torch.float64)
    ln = torch.nn.LayerNorm(64, eps=1e-6).double()
    with torch.no_grad():
        ln.weight.copy_(1 + 0.3 * torch.randn(64, generator=g, dtype=torch.float64))
        ln.bias.copy_(0.2 * torch.randn(64, generator=g, dtype=torch.float64))
    wf, bf, c = fold_layernorm(w, b, ln)
    assert c.dtype == torch.float32 and c.numel() % 4 == 0            # f32x4 loads in the epilogue
    mean = x.mean(-1, keepdim=True)
    rstd = torch.rsqrt(x.var(-1, unbiased=False, keepdim=True) + ln.eps)
    got = rstd * (x @ wf.t() - mean * c[: w.shape[0]].double()) + bf
    want = ln(x) @ w.t() + b
    assert torch.allclose(got, want, rtol=1e-5, atol=1e-5)            # c is kept in f32 (mean * c carries its rounding)


def test_column_sums_are_taken_of_the_rounded_weight():
    """In bf16 the MFMA multiplies the ROUNDED W' - the mean's share must be computed from exactly those values."""
    g = torch.Generator().manual_seed(2)
    w = (torch.randn(16, 128, generator=g) * 0.05).to(torch.bfloat16)
    b = torch.zeros(16, dtype=torch.bfloat16)
    ln = torch.nn.LayerNorm(128).to(torch.bfloat16)
    with torch.no_grad():
        ln.weight.copy_((1 + 0.3 * torch.randn(128, generator=g)).to(torch.bfloat16))
    wf, _, c = fold_layernorm(w, b, ln)
    assert wf.dtype == torch.bfloat16
    assert torch.equal(c[:16], wf.float().sum(1))


def test_prepared_cache_follows_parameter_versions_and_epochs():
    cache = ops.Prepared()
    p = torch.nn.Parameter(torch.ones(4))
    calls = []

    def make():
        calls.append(1)
        return p.detach() * 2

    a = cache.get("k", [p], make)
    assert cache.get("k", [p], make) is a and len(calls) == 1          # unchanged parameter: cached
    with torch.no_grad():
        p.add_(1)                                                     # in-place update bumps the version counter
    b = cache.get("k", [p], make)
    assert len(calls) == 2 and torch.equal(b, torch.full((4,), 4.0))
    cache.get("k", [p], make, epoch=1)                                # the trainer writes weights through raw pointers: epoch
    assert len(calls) == 3
    cache.get("k", [p], make, epoch=1)
    assert len(calls) == 3


def test_siglip_fold_epoch_only_for_trainable_parameters():
    """aki_amd/siglip.py::_epoch: cached transforms of FROZEN weights (the vision tower of the reference's recipes, src/vlm.py:184-207 runs it under
    no_grad) are not keyed on the trainer's weight epoch - a trainer writes only parameters with requires_grad; trainable ones follow the epoch."""
    from aki_amd import siglip, train_ops as T
    frozen = torch.nn.Parameter(torch.ones(4), requires_grad=False)
    live = torch.nn.Parameter(torch.ones(4))
    e0 = T._EPOCH
    try:
        T.bump_weight_epoch()
        assert siglip._epoch([frozen]) == 0
        assert siglip._epoch([frozen, live]) == T._EPOCH == e0 + 1
        cache, calls = ops.Prepared(), []
        make = lambda: calls.append(1) or frozen.detach() * 2
        cache.get("k", [frozen], make, siglip._epoch([frozen]))
        T.bump_weight_epoch()
        cache.get("k", [frozen], make, siglip._epoch([frozen]))
        assert len(calls) == 1                                            # an optimizer step does not rebuild a frozen weight's transform
        frozen.requires_grad_(True)                                       # un-freezing it does
        cache.get("k", [frozen], make, siglip._epoch([frozen]))
        assert len(calls) == 2
    finally:
        T._EPOCH = e0
