#!/usr/bin/env python3
"""ONE first token of `AKI.generate` (1 image + 512-token prompt, batch 1: vision tower, connector, splice, MMA prefill into the KV cache,
head on the last row, pick) as the GPU saw it.  Run under rocprofv3 by tools/profile_first_token.sh; a device-side marker (a recognisable
fill kernel) brackets the last of N calls so that the summary can cut that call out of the trace.
    python tools/first_token_profile.py [--calls 4] [--fp8]                       (plain: prints host / total ms per call)
    python tools/first_token_profile.py --summarize <trace dir> <out prefix>      (per-phase and per-kernel table of the bracketed call)"""
import argparse, csv, glob, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

MARK = 12345          # the bracket: torch.full of this many float64 elements (nothing in generate fills doubles: the last two such kernels are the bracket)


def phase_of(name: str) -> str:
    n = name
    if "gemm_bf16_kernel" in n:
        return "gemm"
    if "connector" in n or "ln_" in n:
        return "connector"
    if "mma_attn_bf16" in n:
        return "decoder attention"
    if "attn_nc_bf16" in n:
        return "vision attention"
    if "gemv" in n or "skinny" in n:
        return "head (last row)"
    if "row_stats" in n or "norm" in n:
        return "norm / statistics"
    if "splice" in n or "mask" in n or "im2col" in n or "patch" in n:
        return "embed / splice / mask"
    if "greedy_pick" in n:
        return "pick"
    return "other (torch)"


def summarize(src, dst):
    tr = glob.glob(os.path.join(src, "*", "*_kernel_trace.csv"))
    rows = sorted(csv.DictReader(open(tr[0])), key=lambda r: int(r["Start_Timestamp"]))
    fills = [i for i, r in enumerate(rows) if "FillFunctor<double>" in r["Kernel_Name"]]
    # the two LAST float64 fills bracket the profiled call
    a, b = fills[-2], fills[-1]
    call = rows[a + 1:b]
    t0, t1 = int(call[0]["Start_Timestamp"]), int(call[-1]["End_Timestamp"])
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in call)
    by_phase, by_kernel = {}, {}
    for r in call:
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        ph = phase_of(r["Kernel_Name"])
        by_phase.setdefault(ph, [0, 0])
        by_phase[ph][0] += 1
        by_phase[ph][1] += d
        k = r["Kernel_Name"].replace("void ", "")[:110]
        by_kernel.setdefault(k, [0, 0])
        by_kernel[k][0] += 1
        by_kernel[k][1] += d
    with open(dst + "_first_token_trace.txt", "w") as f:
        f.write(f"one first token (prompt 655, batch 1): {len(call)} launches, window {(t1 - t0) / 1e6:.3f} ms, kernels busy {busy / 1e6:.3f} ms, "
                f"idle between kernels {(t1 - t0 - busy) / 1e6:.3f} ms\n\nby phase: launches, ms\n")
        for ph, (n, d) in sorted(by_phase.items(), key=lambda kv: -kv[1][1]):
            f.write(f"  {ph:28s} {n:5d} {d / 1e6:8.3f}\n")
        f.write("\nby kernel: launches, total ms, average us\n")
        for k, (n, d) in sorted(by_kernel.items(), key=lambda kv: -kv[1][1])[:40]:
            f.write(f"  {n:5d} {d / 1e6:8.3f} {d / n / 1e3:8.1f}  {k}\n")
    print(open(dst + "_first_token_trace.txt").read())


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--summarize":
        return summarize(sys.argv[2], sys.argv[3])
    ap = argparse.ArgumentParser()
    ap.add_argument("--calls", type=int, default=4)
    ap.add_argument("--fp8", action="store_true")
    a = ap.parse_args()
    import torch
    import bench
    from aki_amd.factory import build_aki
    dev = torch.device("cuda", 0)
    model = build_aki(dtype=torch.bfloat16, device=dev, seed=0).eval()
    if a.fp8:
        model.lang_model.enable_fp8()
    vx, ids, am = bench.synth_batch(1, dev, torch.bfloat16, model.media_token_id, seed=1000)
    res = []
    for i in range(a.calls + 2):
        torch.cuda.synchronize()
        if i == a.calls + 1:
            torch.full((MARK,), 1.0, dtype=torch.float64, device=dev)
        t0 = time.perf_counter()
        model.generate(vx, ids, attention_mask=am, max_new_tokens=1, do_sample=False, eos_token_id=[])
        th = time.perf_counter() - t0
        if i == a.calls + 1:
            torch.full((MARK,), 2.0, dtype=torch.float64, device=dev)
        torch.cuda.synchronize()
        res.append((round(th * 1e3, 2), round((time.perf_counter() - t0) * 1e3, 2)))
    print(json.dumps({"host_issue_ms, total_ms per call": res}))


if __name__ == "__main__":
    main()
