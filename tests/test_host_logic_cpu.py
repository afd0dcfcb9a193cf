"""Host-side logic of round 5 that needs no GPU: the greedy pick's embedding-table selection, the cheap parameter signature behind the cached
per-layer tables of the one-call layer loops, and the activation-checkpointing switch (on a plain torch module: the recomputation is torch's)."""
import ctypes as C

import pytest
import torch
from torch import nn


def test_embedding_tables_selection():
    from aki_amd.aki import _embedding_tables
    from aki_amd.helpers import DecoupledEmbedding
    bf = torch.bfloat16
    logits = torch.zeros(1, 103, dtype=bf)
    dec = DecoupledEmbedding(max_original_id=99, num_additional_embeddings=3, num_original_embeddings=100, embedding_dim=16, pad_token_id=0).to(bf)
    w, extra, max_orig = _embedding_tables(dec, logits)
    assert w is dec.weight and extra is dec.additional_embedding.weight and max_orig == 99
    assert _embedding_tables(dec, torch.zeros(1, 104, dtype=bf)) is None                 # a logit column without an embedding row
    assert _embedding_tables(dec.float(), logits) is None                                # not bf16: the module's own forward is used
    plain = nn.Embedding(103, 16).to(bf)
    assert _embedding_tables(plain, logits) == (plain.weight, None, 102)
    assert _embedding_tables(nn.Embedding(103, 12).to(bf), logits) is None               # rows of 24 bytes: not 16-byte chunks
    assert _embedding_tables(nn.Linear(4, 4), logits) is None                            # not an embedding at all


def test_params_signature_and_layer_table():
    from aki_amd import _lib, ops
    ps = [nn.Parameter(torch.randn(4, 4)) for _ in range(6)]
    s0 = ops.params_signature(ps)
    assert s0 == ops.params_signature(ps)
    with torch.no_grad():
        ps[3].mul_(2.0)                                      # in place through torch: version counter
    s1 = ops.params_signature(ps)
    assert s1 != s0
    ps[1].data = ps[1].data.clone()                          # re-allocated: another address
    assert ops.params_signature(ps) != s1
    tb = ops.LayerTable(_lib.DecoderLayer)
    rows = [(ps[0], ps[1], ps[2], ps[3], None, None), (ps[4], ps[5], ps[0], ps[1], ps[2], ps[3])]
    arr = tb.get(rows)
    assert len(arr) == 2 and arr[0].w_qkv == ps[0].data_ptr() and arr[0].k_cache is None and arr[1].v_cache == ps[3].data_ptr()
    assert tb.get(rows) is arr                               # same pointers: the array is kept
    rows2 = [rows[0], (ps[4], ps[5], ps[0], ps[1], None, None)]
    arr2 = tb.get(rows2)
    assert arr2 is not arr and arr2[1].k_cache is None and tb.keep is rows2
    assert C.sizeof(_lib.DecoderLayer) == 6 * 8 and C.sizeof(_lib.SiglipLayer) == 10 * 8 and C.sizeof(_lib.PerceiverLayer) == 11 * 8


def test_activation_checkpointing_switch_recomputes_and_keeps_names():
    from aki_amd.vlm import _activate_checkpointing

    class Block(nn.Module):
        def __init__(self):
            super().__init__()
            self.lin = nn.Linear(8, 8)
            self.calls = 0

        def forward(self, x, scale=1.0):
            self.calls += 1
            return torch.tanh(self.lin(x)) * scale

    torch.manual_seed(0)
    blk = Block()
    x = torch.randn(3, 8, requires_grad=True)
    y = blk(x, scale=2.0)
    y.sum().backward()
    g_ref, gx_ref = blk.lin.weight.grad.clone(), x.grad.clone()
    blk.zero_grad()
    x.grad = None
    keys = sorted(blk.state_dict())
    _activate_checkpointing(blk)
    assert sorted(blk.state_dict()) == keys and blk._ckpt_active
    blk.calls = 0
    y2 = blk(x, scale=2.0)
    assert blk.calls == 1
    y2.sum().backward()
    assert blk.calls == 2, "the forward was not run again inside backward"
    assert torch.equal(blk.lin.weight.grad, g_ref) and torch.equal(x.grad, gx_ref) and torch.equal(y2, y)
    blk.calls = 0
    with torch.no_grad():                                    # inference goes straight through
        blk(x)
    assert blk.calls == 1
