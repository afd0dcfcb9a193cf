#!/usr/bin/env python3
"""Idle time between kernels in a rocprofv3 --kernel-trace database (the *_results.db it writes): per forward step (a step =
32 launches of the attention core) wall, busy and idle time, and the idle gaps grouped by the pair of kernels around them.

    python tools/trace_gaps.py gpurun_out/prof/x_results.db [--min-us 2]
"""
import argparse
import collections
import re
import sqlite3


def short(name):
    m = re.search(r"gemm_bf16_kernelILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELb(\d)ELi(\d+)ELb(\d)", name)
    if m:
        return "gemm<" + ",".join(m.groups()) + ">"
    m = re.search(r"\d+aki\d+(\w+?)(I|E|P)", name)
    return m.group(1) if m else name[:40]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--min-us", type=float, default=2.0)
    a = ap.parse_args()
    cur = sqlite3.connect(a.db).cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if "kernel_dispatch" in t][0]
    ks = [t for t in tabs if "kernel_symbol" in t][0]
    rows = list(cur.execute(f"select d.start,d.end,s.kernel_name from {kd} d join {ks} s on d.kernel_id=s.id order by d.start"))
    idx = [i for i, r in enumerate(rows) if "mma_attn_bf16_kernel" in r[2]]
    n_steps = len(idx) // 32
    for s in range(max(0, n_steps - 4), n_steps - 1):
        lo, hi = idx[s * 32], idx[(s + 1) * 32]
        wall = rows[hi][0] - rows[lo][0]
        busy = sum(r[1] - r[0] for r in rows[lo:hi])
        groups = collections.defaultdict(lambda: [0, 0.0])
        for i in range(lo, hi - 1):
            g = (rows[i + 1][0] - rows[i][1]) / 1e3
            if g >= a.min_us:
                k = (short(rows[i][2]), short(rows[i + 1][2]))
                groups[k][0] += 1
                groups[k][1] += g
        print(f"step {s}: wall {wall / 1e6:.3f} ms, busy {busy / 1e6:.3f} ms, idle {(wall - busy) / 1e6:.3f} ms, {hi - lo} kernels")
        if s == n_steps - 2:
            for k, (c, tot) in sorted(groups.items(), key=lambda kv: -kv[1][1])[:14]:
                print(f"   {c:4d} x {tot / c:7.1f} us   {k[0]}  ->  {k[1]}")


if __name__ == "__main__":
    main()
