"""Does splitting the step's batch over two HIP streams (two host threads) beat one stream?  Full AKI-4B forward, B=8."""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from aki_amd.factory import build_aki
dev = torch.device("cuda", 0)
model = build_aki(dtype=torch.bfloat16, device=dev).eval()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
vx, ids, am = bench.synth_batch(B, dev, torch.bfloat16, model.media_token_id, seed=1)


def run_one():
    with torch.no_grad():
        return model(vx, ids, attention_mask=am).logits


def run_split(nsplit):
    streams = [torch.cuda.Stream() for _ in range(nsplit)]
    outs = [None] * nsplit
    cur = torch.cuda.current_stream()
    for s in streams:
        s.wait_stream(cur)
    per = B // nsplit

    def work(i):
        torch.cuda.set_device(dev)
        with torch.no_grad(), torch.cuda.stream(streams[i]):
            sl = slice(i * per, (i + 1) * per)
            outs[i] = model(vx[sl], ids[sl], attention_mask=am[sl]).logits
    th = [threading.Thread(target=work, args=(i,)) for i in range(nsplit)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for s in streams:
        cur.wait_stream(s)
    return outs


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


ref = run_one()
sp = torch.cat(run_split(2), 0)
torch.cuda.synchronize()
print("max |split - single| =", (sp.float() - ref.float()).abs().max().item())
print("single stream %.3f ms" % timeit(run_one))
for k in (2, 4):
    if B % k == 0:
        print("%d streams     %.3f ms" % (k, timeit(lambda: run_split(k))))
