"""Does the SigLIP tower (27 layers of 10-65 us kernels, most of them a single round of workgroups) run faster when the
batch of images is split over two HIP streams, so that the ramp-up / tail of one half's kernels overlaps the other half's?
    python tools/siglip_streams_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aki_amd.siglip import SiglipVisionTransformer, make_siglip_config

dev = "cuda"
torch.manual_seed(0)
vt = SiglipVisionTransformer(make_siglip_config()).to(dev).to(torch.bfloat16).eval()
x = (torch.rand(8, 3, 336, 336, device=dev) * 2 - 1).to(torch.bfloat16)


def one(xx):
    with torch.no_grad():
        return vt(xx, interpolate_pos_encoding=True).last_hidden_state


def split(n):
    cur = torch.cuda.current_stream()
    streams = [torch.cuda.Stream() for _ in range(n)]
    outs = []
    for s_, part in zip(streams, x.chunk(n)):
        s_.wait_stream(cur)
        with torch.cuda.stream(s_):
            outs.append(one(part))
    for s_ in streams:
        cur.wait_stream(s_)
    return torch.cat(outs)


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n):
            fn()
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / n)
    return best


ref = one(x)
for n in (1, 2, 4):
    o = one(x) if n == 1 else split(n)
    t = timeit((lambda: one(x)) if n == 1 else (lambda: split(n)))
    print(f"{n} stream(s): {t:7.3f} ms   max |diff vs one stream| {float((o.float() - ref.float()).abs().max()):.4f}", flush=True)
