import sys, os
sys.path.insert(0, os.getcwd())
import torch
import torch.utils.checkpoint as ckpt
from aki_amd import train_ops as T, ops
DEV, BF = "cuda", torch.bfloat16
g = torch.Generator().manual_seed(0)
rnd = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(BF).to(DEV)

def compare(name, fn, inputs, params):
    res = []
    for use in (False, True):
        ins = [t.detach().clone().requires_grad_(True) for t in inputs]
        ps = [p.detach().clone().requires_grad_(True) for p in params]
        y = ckpt.checkpoint(fn, *ins, *ps, use_reentrant=False) if use else fn(*ins, *ps)
        gy = torch.ones_like(y) * 0.01 + torch.arange(y.numel(), device=DEV).view(y.shape).to(BF) * 1e-6
        (y.float() * gy.float()).sum().backward()
        torch.cuda.synchronize()
        res.append((y.detach().clone(), [t.grad.float().clone() for t in ins + ps]))
    bad = [i for i, (a, b) in enumerate(zip(res[0][1], res[1][1])) if not torch.equal(a, b)]
    rel = [round(float((res[0][1][i] - res[1][1][i]).norm() / (res[0][1][i].norm() + 1e-12)), 4) for i in bad]
    print(f"{name}: forward equal {torch.equal(res[0][0], res[1][0])}; grads differing {bad} rel {rel}", flush=True)

M, d, F = 256, 384, 1024
x, w1, w2 = rnd(M, d), rnd(F, d, sc=0.05), rnd(d, F, sc=0.05)
compare("linear", lambda x, w1: T.linear(x, w1), [x], [w1])
compare("linear-linear", lambda x, w1, w2: T.linear(T.linear(x, w1), w2), [x], [w1, w2])
compare("linear+residual", lambda x, r, w1, w2: T.linear(T.linear(x, w1), w2, None, r), [x, rnd(M, d)], [w1, w2])
nw = (1 + 0.1 * torch.randn(d, generator=g)).to(BF).to(DEV)
compare("rmsnorm", lambda x, nw: T.rmsnorm(x, nw, 1e-5), [x], [nw])
def blk(x, nw, w1, w2):
    xn, hr = T.rmsnorm_residual(x, nw, 1e-5)
    return T.linear(T.linear(xn, w1), w2, None, hr)
compare("rmsnorm_residual block", blk, [x], [nw, w1, w2])
wg = rnd(2 * F, d, sc=0.05)
compare("swiglu mlp", lambda x, wg, w2: T.linear(T.SwigluFn.apply(T.linear(x, wg)), w2), [x], [wg, w2])
compare("gelu", lambda x, w1, w2: T.linear(T.GeluFn.apply(T.linear(x, w1)), w2), [x], [w1, w2])
compare("layernorm", lambda x, nw, nb: T.layernorm(x, nw, nb, 1e-5), [x], [nw, (0.1 * torch.randn(d, generator=g)).to(BF).to(DEV)])
# one full decoder layer
from aki_amd.phi3 import Phi3DecoderLayer, make_phi3_config, Phi3RotaryTables
cfg = make_phi3_config(hidden_size=384, intermediate_size=1024, num_attention_heads=4, num_key_value_heads=4, num_hidden_layers=1)
torch.manual_seed(1)
ly = Phi3DecoderLayer(cfg, 0).to(DEV).to(BF).train()
rot = Phi3RotaryTables(cfg)
B, L = 2, 128
cos, sin = rot.tables(L, DEV)
table = ops.MaskTable.causal(B, L, DEV)
h = rnd(B, L, 384)
for use in (False, True):
    ly.zero_grad()
    hin = h.detach().clone().requires_grad_(True)
    out = ckpt.checkpoint(ly, hin, cos, sin, table, None, None, use_reentrant=False) if use else ly(hin, cos, sin, table, None, None)
    out.float().pow(2).sum().backward()
    torch.cuda.synchronize()
    r = (out.detach().clone(), hin.grad.float().clone(), {n: p.grad.float().clone() for n, p in ly.named_parameters()})
    if not use:
        base = r
    else:
        print("decoder layer: forward equal", torch.equal(base[0], r[0]), "dh equal", torch.equal(base[1], r[1]),
              {n: torch.equal(base[2][n], r[2][n]) for n in base[2]})
