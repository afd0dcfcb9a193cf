#!/usr/bin/env python3
"""Where the time of a small-M GEMM launch goes (lab library; workgroup `--block` stamps its lifetime, prologue, K loop and epilogue in
shader cycles): o_proj / down at M = 655 under a few (variant, ksplit) settings, cold (12 rotating operand sets) and warm (one set).
    python tools/small_m_probe.py [--M 655] [--configs -1:1,0:1,2:1,0:3,3:3] [--block 0]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aki_amd import _lib, ops

dev = "cuda"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--M", type=int, default=655)
    ap.add_argument("--configs", default="-1:1,0:1,2:1,0:3,3:3")
    ap.add_argument("--block", type=int, default=0)
    a = ap.parse_args()
    lib = _lib.load_lab()
    _lib._lib = lib
    ops.SPLITK_WS_MIN_BYTES = 512 << 20
    g = torch.Generator(device=dev).manual_seed(0)
    rnd = lambda *s, sc=1.0: (torch.randn(*s, device=dev, generator=g) * sc).to(torch.bfloat16)
    M, d, F, NB = a.M, 3072, 8192, 12
    x, r, act = [rnd(M, d) for _ in range(NB)], [rnd(M, d) for _ in range(NB)], [rnd(M, F) for _ in range(NB)]
    wo, wd = [rnd(d, d, sc=0.02) for _ in range(NB)], [rnd(d, F, sc=0.02) for _ in range(NB)]
    y = torch.empty(M, d, device=dev, dtype=torch.bfloat16)
    st = ops.new_stats(M, dev)
    probe = torch.zeros(32, dtype=torch.int64, device=dev)
    lib.aki_lab_set_clock_probe(probe.data_ptr())
    lib.aki_lab_set_probe_block(a.block)
    cases = {"o_proj K3072": lambda i: ops.linear(x[i], wo[i], residual=r[i], stats_out=st, stats_eps=1e-5, out=y),
             "o_proj plain": lambda i: ops.linear(x[i], wo[i], out=y),
             "down K8192": lambda i: ops.linear(act[i], wd[i], residual=r[i], stats_out=st, stats_eps=1e-5, out=y)}
    print(f"# M={M}; workgroup {a.block}; columns: launch us (events) | wg lifetime us | MHz | prologue, K loop, epilogue kcycles | K-loop cycles per step")
    for name, fn in cases.items():
        nk = (8192 if "down" in name else 3072) // 64
        for cfg in a.configs.split(","):
            v, ks = [int(t_) for t_ in cfg.split(":")]
            lib.aki_lab_set_small_m(v, ks)
            for label, nb in (("cold", NB), ("warm", 1)):
                for i in range(nb):
                    fn(i)
                torch.cuda.synchronize()
                us, rows = [], []
                for rep in range(12):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    fn(rep % nb)
                    e1.record()
                    torch.cuda.synchronize()
                    us.append(e0.elapsed_time(e1) * 1e3)
                    rows.append(probe.tolist())
                # throughput form too: back-to-back launches
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for k in range(48):
                    fn(k % nb)
                e1.record()
                torch.cuda.synchronize()
                b2b = e0.elapsed_time(e1) / 48 * 1e3
                us.sort()
                pr = rows[len(rows) // 2]
                life, ticks, pro, epi = pr[0], pr[1], pr[18], pr[19]
                loop = life - pro - epi
                steps = max(1, nk // max(ks, 1))
                print(f"{name:13s} v{v} k{ks} {label}: back-to-back {b2b:6.1f} us | single {us[len(us)//2]:6.1f} us | wg {ticks / 100:6.1f} us | {life / max(ticks, 1) * 100:5.0f} MHz | "
                      f"{pro / 1e3:5.1f} {loop / 1e3:6.1f} {epi / 1e3:5.1f} | {loop / steps:6.0f}", flush=True)
    lib.aki_lab_set_small_m(-1, 1)
    lib.aki_lab_set_clock_probe(None)


if __name__ == "__main__":
    main()
