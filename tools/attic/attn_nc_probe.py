import sys; sys.path.insert(0, "/root/repo")
import torch
from aki_amd import ops
dev = "cuda"
def loop_us(fn, iters=30):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); torch.cuda.synchronize()
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3
for (B, H, L, D) in ((8, 16, 512, 72), (8, 16, 576, 72), (8, 16, 640, 72), (8, 16, 576, 64), (8, 16, 576, 96), (4, 16, 576, 72), (16, 16, 576, 72)):
    qkv = torch.randn(B, L, 3, H, D, device=dev).to(torch.bfloat16)
    f = lambda: ops.attention(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], D ** -0.5)
    t = sorted(loop_us(f) for _ in range(5))[2]
    fl = 4.0 * B * H * L * L * D
    print(f"B{B} H{H} L{L} D{D}: {t:7.1f} us  {fl / t / 1e6:7.1f} TF/s   workgroups {B * H * ((L + 127) // 128)}")
