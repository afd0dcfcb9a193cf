"""Build libaki_mi355x.so (HIP, gfx950 only) in-tree with hipcc.

    python -m aki_amd.build [--force] [--save-temps] [--no-lab]

The .so lands in aki_amd/lib/ (git-ignored, but it travels to the GPU box with the snapshot).
Cross-compiles without a GPU.  Also builds the oracle's C restatement (oracle/Makefile) when asked
by __graft_entry__.build().
"""
from __future__ import annotations

import argparse
import concurrent.futures as cf
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIBDIR = os.path.join(PKG, "lib")
LIB = os.path.join(LIBDIR, "libaki_mi355x.so")
# Lab twin: the same sources with -DAKI_LAB_HOOKS (adds aki_lab_set_gemm_tile, a process-global tile-forcing switch that the
# product library must not carry).  Used by the forced-tile GEMM tests and tools/siglip_gemm_bench.py only.
LAB_LIB = os.path.join(LIBDIR, "libaki_mi355x_lab.so")
LAB_SOURCES = ["api.hip", "gemm_bf16.hip", "gemm_tn_bf16.hip", "mma_attn_bf16.hip", "mma_attn64_bf16.hip", "decode_chain.hip"]
LAB_ONLY_SOURCES = []     # experiments that exist in the lab library only (none at present)
SOURCES = ["api.hip", "gemm_bf16.hip", "gemm_tn_bf16.hip", "mma_attn_bf16.hip", "mma_attn64_bf16.hip", "attn_nc_bf16.hip", "decode.hip", "decode_chain.hip", "train_kernels.hip", "attn_bwd_bf16.hip", "fp8_quant.hip", "simple_f32.hip", "aux_kernels.hip", "stack.hip"]
ARCH = "gfx950"
# per-file flags: the 64-row attention core places single VALU instructions in MFMA gaps by hand - SLP vectorisation turns its
# f32 adds into v_pk_add_f32 plus moves (MI355X guide: packed f32 VALU beside MFMAs is an anti-lever)
FILE_FLAGS = {"mma_attn64_bf16.hip": ["-fno-slp-vectorize"]}


def csrc_hash() -> str:
    """sha256 over the kernel sources and headers (names + contents, sorted): profiles/*_pmc_summary.json carry the hash of the
    tree their counters were collected on, and bench.py marks `roofline.traffic` stale when it differs from the running tree's."""
    import hashlib
    h = hashlib.sha256()
    files = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))) + [os.path.join(ROOT, "include", "aki_mi355x.h")]
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _newer(target: str, deps) -> bool:
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(d) <= t for d in deps)


def build(force: bool = False, save_temps: bool = False, verbose: bool = True, lab: bool = True) -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(LIBDIR, "obj")
    os.makedirs(objdir, exist_ok=True)
    headers = [os.path.join(CSRC, "aki_device.h"), os.path.join(CSRC, "attn_mma_common.h"), os.path.join(ROOT, "include", "aki_mi355x.h")]
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    lab_only = [os.path.join(CSRC, s_) for s_ in LAB_ONLY_SOURCES]
    if not force and _newer(LIB, srcs + headers) and (not lab or _newer(LAB_LIB, srcs + lab_only + headers)):
        return LIB
    flags = [f"--offload-arch={ARCH}", "-O3", "-fPIC", "-std=c++17", "-I", os.path.join(ROOT, "include"), "-I", CSRC,
             "-Wno-unused-result", "-ffp-contract=off"]
    if save_temps:
        flags += ["-save-temps=obj", "-Rpass-analysis=kernel-resource-usage"]

    def cc(job):
        src, is_lab = job
        obj = os.path.join(objdir, os.path.basename(src).replace(".hip", ".lab.o" if is_lab else ".o"))
        if not force and _newer(obj, [src] + headers):
            return obj, ""
        extra = (["-DAKI_LAB_HOOKS"] if is_lab else []) + FILE_FLAGS.get(os.path.basename(src), [])
        r = subprocess.run([_hipcc()] + flags + extra + ["-c", src, "-o", obj], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
        return obj, r.stderr

    jobs = [(s_, False) for s_ in srcs] + ([(os.path.join(CSRC, s_), True) for s_ in LAB_SOURCES + LAB_ONLY_SOURCES] if lab else [])
    with cf.ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
        results = list(ex.map(cc, jobs))
    objs = [o for o, _ in results[:len(srcs)]]
    lab_objs = [o for o, _ in results[len(srcs):]]
    if verbose:
        for _, log in results:
            if log.strip():
                sys.stderr.write(log)
    r = subprocess.run([_hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB] + objs, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    if lab:
        swapped = {os.path.basename(o).replace(".lab.o", ".o"): o for o in lab_objs}
        mix = [swapped.pop(os.path.basename(o), o) for o in objs] + list(swapped.values())   # + the lab-only objects
        r = subprocess.run([_hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LAB_LIB] + mix, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link (lab) failed:\n{r.stdout}\n{r.stderr}")
    return LIB


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--save-temps", action="store_true")
    ap.add_argument("--no-lab", action="store_true", help="skip libaki_mi355x_lab.so")
    a = ap.parse_args()
    print(build(force=a.force, save_temps=a.save_temps, lab=not a.no_lab))
