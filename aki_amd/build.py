"""Build libaki_mi355x.so (HIP, gfx950 only) in-tree with hipcc.

    python -m aki_amd.build [--force] [--save-temps]

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
SOURCES = ["api.hip", "gemm_bf16.hip", "mma_attn_bf16.hip", "attn_nc_bf16.hip", "decode.hip", "train_kernels.hip", "attn_bwd_bf16.hip", "fp8_quant.hip", "simple_f32.hip", "aux_kernels.hip"]
ARCH = "gfx950"


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


def build(force: bool = False, save_temps: bool = False, verbose: bool = True) -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(LIBDIR, "obj")
    os.makedirs(objdir, exist_ok=True)
    headers = [os.path.join(CSRC, "aki_device.h"), os.path.join(ROOT, "include", "aki_mi355x.h")]
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    if not force and _newer(LIB, srcs + headers):
        return LIB
    flags = [f"--offload-arch={ARCH}", "-O3", "-fPIC", "-std=c++17", "-I", os.path.join(ROOT, "include"), "-I", CSRC,
             "-Wno-unused-result", "-ffp-contract=off"]
    if save_temps:
        flags += ["-save-temps=obj", "-Rpass-analysis=kernel-resource-usage"]

    def cc(src):
        obj = os.path.join(objdir, os.path.basename(src).replace(".hip", ".o"))
        if not force and _newer(obj, [src] + headers):
            return obj, ""
        r = subprocess.run([_hipcc()] + flags + ["-c", src, "-o", obj], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
        return obj, r.stderr

    with cf.ThreadPoolExecutor(max_workers=min(4, len(srcs))) as ex:
        results = list(ex.map(cc, srcs))
    objs = [o for o, _ in results]
    if verbose:
        for _, log in results:
            if log.strip():
                sys.stderr.write(log)
    r = subprocess.run([_hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB] + objs, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--save-temps", action="store_true")
    a = ap.parse_args()
    print(build(force=a.force, save_temps=a.save_temps))
