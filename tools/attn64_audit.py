"""Audit of the 64-row attention core's code object (mma_attn64_bf16.hip names accumulator registers a[56:255] literally in its asm
statements: nothing the COMPILER emits may touch them (it parks spilled VGPRs in a0 upwards), spill to scratch, or sit between an asm LDS read and its wait).

    python tools/attn64_audit.py [--keep DIR]

Compiles the file with -save-temps, then for every mma_attn64 kernel in the .s reports: register counts, scratch, compiler-emitted
v_accvgpr_* (outside ;;#ASMSTART/;;#ASMEND), v_readlane/v_writelane (SGPR spills), and the instruction mix of the tile loop (the
innermost loop that holds an s_barrier).  Exit code 1 when a hard rule is broken."""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "aki_amd", "csrc", "mma_attn64_bf16.hip")


def main():
    rc = audit([])
    if "--lab" in sys.argv:
        print("-- lab build (-DAKI_LAB_HOOKS): the two kernels that must be right there too")
        rc |= audit(["-DAKI_LAB_HOOKS"], only=("ILi0ELi0E", "ILi8ELi0E", "ILi8ELi64E"))
    return rc


def audit(extra, only=None):
    keep = sys.argv[sys.argv.index("--keep") + 1] if "--keep" in sys.argv else None
    d = keep or tempfile.mkdtemp(prefix="attn64_audit_")
    os.makedirs(d, exist_ok=True)
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "aki_amd", "csrc"),
           "-ffp-contract=off", "-fno-slp-vectorize", "-save-temps=obj"] + extra + ["-c", SRC, "-o", os.path.join(d, "a64.o")]
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=d)
    if r.returncode != 0:
        print(r.stderr)
        return 2
    spath = os.path.join(d, [f for f in os.listdir(d) if f.endswith("gfx950.s")][0])
    s = open(spath).read().split("\n")
    bad = 0
    i = 0
    while i < len(s):
        m = re.match(r"^(_ZN3aki22mma_attn64_bf16_kernel\w+):", s[i])
        if not m:
            i += 1
            continue
        name = m.group(1)
        if only and not any(o in name for o in only):
            i += 1
            continue
        j = i
        while not s[j].strip().startswith("s_endpgm"):
            j += 1
        body = s[i:j + 1]
        # the descriptor that follows
        desc = {}
        k = j
        while k < len(s) and ".end_amdhsa_kernel" not in s[k]:
            mm = re.match(r"\s*\.amdhsa_(\w+)\s+(\S+)", s[k])
            if mm:
                desc[mm.group(1)] = mm.group(2)
            k += 1
        in_asm = False
        comp_acc, lanes, scratch, n_mfma, movs, hi_acc = 0, 0, 0, 0, 0, -1
        for ln in body:
            t = ln.strip()
            if t.startswith(";;#ASMSTART"):
                in_asm = True
            elif t.startswith(";;#ASMEND"):
                in_asm = False
            elif not in_asm:
                if t.startswith("v_accvgpr"):
                    comp_acc += 1
                    for mm in re.finditer(r"\ba(\d+)\b", t):
                        hi_acc = max(hi_acc, int(mm.group(1)))
                if t.startswith(("v_readlane", "v_writelane")):
                    lanes += 1
                if t.startswith("scratch_"):
                    scratch += 1
                if t.startswith("v_mov_b32"):
                    movs += 1
            if t.startswith("v_mfma"):
                n_mfma += 1
        hz = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "attn64_hazards.py"), spath, "--kernel", name[len("_ZN3aki22mma_attn64_bf16_kernel"):name.index("EEv")]],
                            capture_output=True, text=True)
        hz_line = hz.stdout.strip().split("\n")[-1] if hz.stdout.strip() else hz.stderr.strip()[-200:]
        if hz.returncode != 0:
            bad = 1
            print(hz.stdout)
        print(f"{name}: {len(body)} lines, {n_mfma} MFMAs, next_free_vgpr {desc.get('next_free_vgpr')}, accum_offset {desc.get('accum_offset')}, "
              f"private_segment {desc.get('private_segment_fixed_size')}; compiler v_accvgpr {comp_acc} (highest a{hi_acc}), lane spills {lanes}, scratch ops {scratch}, v_mov {movs}; hazards: {hz_line}")
        # the exact build (THR = 0) keeps its row sums in VGPRs and names a[64:255]; the product builds also name a[56:63] (row sums)
        limit = 64 if "ILi0E" in name[:name.index("EEv")].split("kernel")[1][:6] else 56
        if hi_acc >= limit or scratch or desc.get("private_segment_fixed_size", "0") != "0":
            bad = 1
        # tile loop: the loop body between the label that precedes the first s_barrier inside a backward branch ... keep it simple:
        # count per basic block that contains MFMAs
        blk, stats = None, {}
        for ln in body:
            t = ln.strip()
            mm = re.match(r"^(\.LBB\d+_\d+):", ln)
            if mm:
                blk = mm.group(1)
                stats[blk] = {"mfma": 0, "valu": 0, "salu": 0, "ds": 0, "vmem": 0, "lane": 0, "mov": 0, "wait": 0, "nop": 0}
                continue
            if blk is None or not t or t.startswith((";", ".")):
                continue
            st = stats[blk]
            if t.startswith("v_mfma"): st["mfma"] += 1
            elif t.startswith(("v_readlane", "v_writelane")): st["lane"] += 1
            elif t.startswith("v_mov"): st["mov"] += 1; st["valu"] += 1
            elif t.startswith("v_"): st["valu"] += 1
            elif t.startswith("s_waitcnt"): st["wait"] += 1
            elif t.startswith("s_nop"): st["nop"] += 1
            elif t.startswith("s_"): st["salu"] += 1
            elif t.startswith("ds_"): st["ds"] += 1
            elif t.startswith(("global_", "buffer_")): st["vmem"] += 1
        for b_, st in stats.items():
            if st["mfma"] >= 6:
                print(f"   {b_}: " + " ".join(f"{k_}={v_}" for k_, v_ in st.items()))
        i = j + 1
    return bad


if __name__ == "__main__":
    sys.exit(main())
