"""Scan the 64-row attention core's code object for the hazards hipcc cannot see (its MFMAs are inline asm: the compiler pads nothing
around them and gives their dead operand registers away at once):
  RAW  a non-MFMA read of a VGPR that an MFMA issued fewer than R wait states earlier writes (LLVM: XDL 8-pass write -> VALU read 11);
  WAR  a VALU write to a VGPR that an MFMA issued fewer than W wait states earlier reads as SrcC (LLVM: 32x32 SrcC read -> VALU write 15;
       A and B are read at issue - LLVM knows no hazard for them).
Wait states are counted the way LLVM's hazard recognizer does: one per instruction, N + 1 for s_nop N.  Straight-line only: a branch or
barrier ends the window (what follows a barrier is far enough; the loop back edge leads to one).
    python tools/attn64_hazards.py [file.s] [--kernel SUBSTR] [-w 6] [-r 14]"""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def regs_of(tok):
    tok = tok.strip().rstrip(",")
    m = re.match(r"^-?\|?v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"^-?\|?v(\d+)\b", tok)
    if m:
        return {int(m.group(1))}
    return set()


def main():
    skip = {i + 1 for i, a in enumerate(sys.argv) if a in ("-w", "-r", "--kernel")}
    args = [a for i, a in enumerate(sys.argv) if i >= 1 and i not in skip and not a.startswith("-")]
    W = int(sys.argv[sys.argv.index("-w") + 1]) if "-w" in sys.argv else 16
    R = int(sys.argv[sys.argv.index("-r") + 1]) if "-r" in sys.argv else 12
    ksub = sys.argv[sys.argv.index("--kernel") + 1] if "--kernel" in sys.argv else "ILi8ELi0E"
    if args:
        path = args[0]
    else:
        d = tempfile.mkdtemp(prefix="attn64_hz_")
        src = os.path.join(ROOT, "aki_amd", "csrc", "mma_attn64_bf16.hip")
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "aki_amd", "csrc"),
                        "-ffp-contract=off", "-fno-slp-vectorize", "-save-temps=obj", "-c", src, "-o", os.path.join(d, "a.o")], check=True, cwd=d, capture_output=True)
        path = os.path.join(d, [f for f in os.listdir(d) if f.endswith("gfx950.s")][0])
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_ZN3aki22mma_attn64_bf16_kernel", l) and ksub in l and ":" in l and not l.startswith("\t"))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    ins = []
    for i in range(start, end):
        t = lines[i].strip()
        if not t or t[0] in ";." or t.endswith(":"):
            continue
        ins.append((i - start, t))
    nwar = nraw = 0

    def states(t):          # wait states an instruction contributes (LLVM counts instructions; s_nop N is N + 1)
        m = re.match(r"s_nop (\d+)", t)
        return int(m.group(1)) + 1 if m else 1

    for k, (ln, t) in enumerate(ins):
        if not t.startswith("v_mfma"):
            continue
        ops = [o.strip() for o in t.split(None, 1)[1].split(",")]
        dst = regs_of(ops[0])
        srcc = regs_of(ops[3]) if len(ops) > 3 else set()
        # WAR on SrcC (the operand an MFMA keeps reading while it runs; A and B are read at issue): a VALU write within W wait states
        ws = 0
        for (ln2, t2) in ins[k + 1:]:
            if ws >= W or t2.startswith(("s_cbranch", "s_branch", "s_barrier", "s_endpgm")):
                break
            if t2.startswith("v_") and not t2.startswith(("v_mfma", "v_cmp", "v_readlane", "v_readfirstlane", "v_accvgpr_write")):
                d2 = regs_of(t2.split(None, 1)[1].split(",")[0])
                hit = d2 & (srcc - dst)
                if hit:
                    nwar += 1
                    print(f"WAR  +{ln}: {t}\n     +{ln2}: {t2}   (writes SrcC v{sorted(hit)} after {ws} wait states)")
            ws += states(t2)
        # RAW: anything but an MFMA reading the result within R wait states
        if dst:
            ws = 0
            for (ln2, t2) in ins[k + 1:]:
                if ws >= R or t2.startswith(("s_cbranch", "s_branch", "s_barrier", "s_endpgm")):
                    break
                if t2.startswith(("v_", "ds_write", "global_store", "buffer_store")) and not t2.startswith("v_mfma"):
                    parts = t2.split(None, 1)
                    ops2 = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
                    rd = set()
                    for o in (ops2 if t2.startswith(("ds_write", "global_store", "buffer_store", "v_accvgpr_write", "v_cmp")) else ops2[1:]):
                        rd |= regs_of(o)
                    hit = rd & dst
                    if hit:
                        nraw += 1
                        print(f"RAW  +{ln}: {t}\n     +{ln2}: {t2}   (reads v{sorted(hit)[:4]} after {ws} wait states)")
                        break
                ws += states(t2)
    print(f"{nwar} WAR, {nraw} RAW candidates (windows {W} / {R} wait states) in {ksub}")
    return 1 if (nwar or nraw) else 0


if __name__ == "__main__":
    sys.exit(main())
