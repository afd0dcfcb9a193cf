"""Audit of the pattern behind round 3's statistics-fold fault: a value produced by a compiler-PAIRED LDS read
(`ds_read2_b32/_b64`, `ds_read2st64_*`) that is first consumed behind a COUNTED `s_waitcnt lgkmcnt(k)`, k > 0 - i.e. the compiler
decided that k younger LDS operations may still be in flight when the paired read's registers are used.  The ISA guarantees in-order
return of LDS operations, so such a wait is legal; the fault of round 3 showed one site where the SUM half of a `ds_read2st64_b64` pair
was consumed early all the same (cause not established).  This tool lists every such site per kernel from the device assembly:

    for f in aki_amd/csrc/*.hip; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I aki_amd/csrc -ffp-contract=off \\
        -S --cuda-device-only $f -o /tmp/s/$(basename $f .hip).s; done
    python tools/lgkm_audit.py /tmp/s/*.s

A site = a `ds_read2*` whose destination registers are read by a later instruction of the same basic block with at least one counted
(k > 0) and no full (k = 0) lgkm wait in between.  Hand-written asm blocks (between ;;#ASMSTART / ;;#ASMEND) are reported separately."""
import re, sys, collections

RD2 = re.compile(r"^\s*(ds_read2(?:st64)?_b(?:32|64))\s+(v\[(\d+):(\d+)\])")
WAIT = re.compile(r"s_waitcnt.*lgkmcnt\((\d+)\)")
REG = re.compile(r"\bv(\d+)\b|v\[(\d+):(\d+)\]")


def regs_of(operands):
    out = set()
    for m in REG.finditer(operands):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def audit(path):
    kernel, in_asm = None, False
    sites = collections.Counter()
    pending = []          # [dest regs, counted-wait seen, opcode, in_asm]
    total = collections.Counter()
    for ln in open(path, errors="replace"):
        s = ln.strip()
        if re.match(r"^[A-Za-z_][\w.$]*:", s) and not s.startswith(".L"):
            kernel = s.split(":")[0]
            pending = []                                   # function label: new kernel
            continue
        if s.startswith(".LBB") or s.startswith("; %bb"):
            pending = []
            continue
        if "#ASMSTART" in s:
            in_asm = True
            continue
        if "#ASMEND" in s:
            in_asm = False
            continue
        if not s or s.startswith(";") or s.startswith("."):
            continue
        m = RD2.match(s)
        w = WAIT.search(s)
        if w:
            k = int(w.group(1))
            if k == 0:
                pending = []
            else:
                for p in pending:
                    p[1] = True
            continue
        if s.startswith("s_waitcnt") and "lgkmcnt" not in s:
            continue
        # a consumer?
        ops = s.split(None, 1)[1] if " " in s else ""
        used = regs_of(ops.split(",", 1)[1]) if (m is None and "," in ops) else (regs_of(ops) if m is None else set())
        keep = []
        for p in pending:
            if used & p[0]:
                if p[1]:
                    sites[(kernel, p[2], "asm" if p[3] else "compiler")] += 1
                # consumed (with or without a counted wait): done with it
            else:
                keep.append(p)
        pending = keep
        if m:
            total[(kernel, m.group(1))] += 1
            pending.append([set(range(int(m.group(3)), int(m.group(4)) + 1)), False, m.group(1), in_asm])
        # writes to a pending destination by another instruction end its life
    return sites, total


def main():
    grand = 0
    for path in sys.argv[1:]:
        sites, total = audit(path)
        n2 = sum(total.values())
        print(f"== {path.split('/')[-1]}: {n2} paired LDS reads, {sum(sites.values())} consumed behind a counted lgkmcnt(k>0)")
        for (kernel, op, who), n in sorted(sites.items(), key=lambda t: -t[1])[:40]:
            print(f"   {n:5d}  {op:18s} {who:8s} {str(kernel)[:130]}")
        grand += sum(sites.values())
    print(f"total sites: {grand}")


if __name__ == "__main__":
    main()
