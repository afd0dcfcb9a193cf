"""CPU-side checks that need no GPU: the C-ABI library loads and exports every symbol include/aki_mi355x.h declares,
host-side argument validation answers without launching anything, and the oracle's C restatement of the integer
path agrees with the golden vectors."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import ROOT, load_golden
from golden import gen


def header_symbols():
    src = open(os.path.join(ROOT, "include", "aki_mi355x.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(aki_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as ge
    ge.build()                                    # cross-compiles for gfx950 without a GPU
    from aki_amd import _lib
    return _lib.load()


def test_library_exports_every_declared_symbol(lib):
    from aki_amd import _lib
    declared = header_symbols()
    assert len(declared) >= 18
    assert sorted(_lib.SIGNATURES) == declared, "ctypes table and header disagree"
    for name in declared:
        assert hasattr(lib, name), f"{name} is declared in include/aki_mi355x.h but not exported"
    assert lib.aki_abi_version() == 17
    exported = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    assert "debug" not in exported and "aki_lab_" not in exported, "lab / debug hooks must not ship in the product library"
    assert b"aligned" in lib.aki_strerror(-3)


STRUCTS = {"aki_mma_rect": "MmaRect", "aki_mma_attn_core_args": "MmaAttnCoreArgs", "aki_mma_attn_args": "MmaAttnArgs",
           "aki_attn_args": "AttnArgs", "aki_linear_args": "LinearArgs", "aki_splice_args": "SpliceArgs",
           "aki_attn_bwd_args": "AttnBwdArgs", "aki_decode_chain_layer": "DecodeChainLayer", "aki_decode_chain_args": "DecodeChainArgs",
           "aki_decoder_layer": "DecoderLayer", "aki_decoder_stack_args": "DecoderStackArgs", "aki_siglip_layer": "SiglipLayer",
           "aki_siglip_stack_args": "SiglipStackArgs", "aki_perceiver_layer": "PerceiverLayer", "aki_perceiver_stack_args": "PerceiverStackArgs"}


def header_structs():
    """{struct name: [field names in declaration order]} parsed out of include/aki_mi355x.h."""
    src = open(os.path.join(ROOT, "include", "aki_mi355x.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    out = {}
    for body, name in re.findall(r"typedef\s+struct\s*\{(.*?)\}\s*(\w+)\s*;", src, flags=re.S):
        fields = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            for part in decl.split(","):
                fields.append(re.findall(r"(\w+)\s*$", part.strip())[0])
        out[name] = fields
    return out


def c_layouts(tmp_path):
    """sizeof / offsetof of every args struct as the C compiler sees the header: {struct: (size, {field: (offset, size)})}."""
    structs = header_structs()
    lines = ["#include <stdio.h>", "#include <stddef.h>", '#include "aki_mi355x.h"', "int main(void) {"]
    for sname, fields in structs.items():
        lines.append(f'  printf("S {sname} %zu\\n", sizeof({sname}));')
        for f in fields:
            lines.append(f'  printf("F {sname} {f} %zu %zu\\n", offsetof({sname}, {f}), sizeof((({sname}*)0)->{f}));')
    lines += ["  return 0;", "}"]
    csrc = tmp_path / "layout.c"
    csrc.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c11", "-I", os.path.join(ROOT, "include"), str(csrc), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout
    lay = {}
    for ln in out.splitlines():
        t = ln.split()
        if t[0] == "S":
            lay[t[1]] = (int(t[2]), {})
        else:
            lay[t[1]][1][t[2]] = (int(t[3]), int(t[4]))
    return structs, lay


def assert_same_layout(cls, sname, fields, lay):
    size, offs = lay[sname]
    assert [f for f, _ in cls._fields_] == fields, f"{cls.__name__}: field names / order differ from {sname}"
    assert C.sizeof(cls) == size, f"{cls.__name__}: {C.sizeof(cls)} bytes, header says {size}"
    for f, ct in cls._fields_:
        d = getattr(cls, f)
        assert (d.offset, d.size) == offs[f], f"{cls.__name__}.{f}: ctypes (offset, size) {(d.offset, d.size)} vs C {offs[f]}"


def integration_md_binding():
    """The python block of INTEGRATION.md section B, verbatim."""
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blk = md[md.index("<!-- BEGIN mi355x_binding.py -->"):md.index("<!-- END mi355x_binding.py -->")]
    return blk[blk.index("```python") + len("```python"):blk.rindex("```")]


def test_struct_layouts_match_the_header(lib, tmp_path, monkeypatch):
    """Every args struct: the ctypes mirror in aki_amd/_lib.py AND the one a maintainer would paste from INTEGRATION.md
    against sizeof/offsetof of a C program compiled from the header (round-1 shipped a stale 128-byte MmaAttnArgs in the
    document while the library read kv_capacity at offset 128)."""
    from aki_amd import _lib
    structs, lay = c_layouts(tmp_path)
    assert set(structs) == set(STRUCTS), "a struct was added to the header: mirror it in _lib.py and list it here"
    for sname, cname in STRUCTS.items():
        assert_same_layout(getattr(_lib, cname), sname, structs[sname], lay)
    assert lay["aki_mma_attn_args"][0] == 160 and lay["aki_mma_attn_args"][1]["kv_capacity"][0] == 128
    monkeypatch.setenv("AKI_MI355X_SO", _lib.LIB_PATH)
    ns = {}
    exec(compile(integration_md_binding(), "INTEGRATION.md#B", "exec"), ns)     # loads the library, binds the prototypes
    assert_same_layout(ns["MmaAttnArgs"], "aki_mma_attn_args", structs["aki_mma_attn_args"], lay)
    for fn, (res, args) in _lib.SIGNATURES.items():          # prototypes the document binds must agree with _lib.py
        bound = getattr(ns["_lib"], fn)
        if bound.argtypes is not None:
            assert len(bound.argtypes) == len(args), fn
            assert [C.sizeof(a) for a in bound.argtypes] == [C.sizeof(a) for a in args], fn


def test_host_side_validation_without_gpu(lib):
    from aki_amd import _lib as L
    a = L.LinearArgs()                                            # all-null arguments
    assert lib.aki_linear_fwd(C.byref(a), None) == -1             # AKI_ERR_INVALID_ARG, nothing launched
    c = L.MmaAttnCoreArgs()
    assert lib.aki_mma_attn_core_fwd(C.byref(c), None, 0, None) == -1
    assert lib.aki_mma_attn_workspace_bytes(8, 32, 655, 96, 0) >= 3 * 8 * 32 * 655 * 96 * 2
    assert lib.aki_patch_embed_workspace_bytes(8, 336, 14, 0) >= 8 * 576 * 640 * 2
    # round 5: the one-call layer loops validate on the host before anything is launched
    assert lib.aki_decoder_stack_fwd(C.byref(L.DecoderStackArgs()), None) == -1
    assert lib.aki_siglip_stack_fwd(C.byref(L.SiglipStackArgs()), None) == -1
    assert lib.aki_perceiver_stack_fwd(C.byref(L.PerceiverStackArgs()), None) == -1
    # the batched decode chain's workspace: grows with the batch; batch <= 1 is the one-sequence layout; more than eight sequences are refused
    one = lib.aki_decode_chain_workspace_bytes(32, 3072, 32, 8192, 800)
    assert lib.aki_decode_chain_batch_workspace_bytes(32, 3072, 32, 8192, 800, 1) == one and lib.aki_decode_chain_batch_workspace_bytes(32, 3072, 32, 8192, 800, 0) == one
    assert lib.aki_decode_chain_batch_workspace_bytes(32, 3072, 32, 8192, 800, 8) > lib.aki_decode_chain_batch_workspace_bytes(32, 3072, 32, 8192, 800, 2) > 0
    assert lib.aki_decode_chain_batch_workspace_bytes(32, 3072, 32, 8192, 800, 9) == 0
    assert lib.aki_decode_chain_batch_error_offset(32, 32, 8) > lib.aki_decode_chain_error_offset(32, 32) > 0
    need = lib.aki_decoder_stack_workspace_bytes(1, 32, 655, 96, 3072, 8192, 1)
    assert need >= 655 * (2 * 3072 * 2 + 3072 * 2 + 3072 * 2 + 8192 * 2) and need % 256 == 0       # q + o + two residual streams + SwiGLU output
    assert lib.aki_decoder_stack_workspace_bytes(1, 32, 655, 96, 3072, 8192, 0) > need                 # + k, v when no KV cache takes them
    assert lib.aki_siglip_stack_workspace_bytes(1, 576, 1152, 16) >= 576 * 1152 * 2 * 6
    assert lib.aki_perceiver_stack_workspace_bytes(576, 144, 1152, 8, 64, 4608) >= (576 + 144) * 1152 * 2
    # the split-K planner: a one-sample prefill's N = 3072 GEMMs are split (a workspace is wanted), the headline batch's are not
    assert lib.aki_linear_splitk_workspace_bytes(655, 3072, 3072) > 0 and lib.aki_linear_splitk_workspace_bytes(655, 3072, 8192) > 0
    assert lib.aki_linear_splitk_workspace_bytes(207, 3072, 8192) > 0
    assert lib.aki_linear_splitk_workspace_bytes(5240, 3072, 8192) == 0 and lib.aki_linear_splitk_workspace_bytes(655, 16384, 3072) == 0
    assert lib.aki_linear_splitk_workspace_bytes(655, 3072, 1152) == 0                                # 18 K-steps: nothing to split


def test_product_package_never_imports_the_oracle():
    """The product path must not route through the oracle (or any CPU fallback)."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "aki_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "aki_oracle" not in txt and "oracle/" not in txt.replace("oracle/Makefile", ""), f


def test_missing_library_fails_loudly(tmp_path, monkeypatch):
    from aki_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.AkiError):
        _lib.load()


def test_cpu_tensors_are_refused():
    import torch
    from aki_amd import ops
    with pytest.raises(ops.AkiError):
        ops.rmsnorm(torch.zeros(2, 64), torch.ones(64), 1e-5)


def test_c_restatement_of_mask_and_splice_index():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle")], check=True)
    lib = C.CDLL(os.path.join(ROOT, "oracle", "_build", "liboracle_mask.so"))
    g = load_golden("mask_cases.npz")
    for i, (am, s, t, e) in enumerate(gen.mask_cases()):
        n = len(am)
        out = np.zeros((n, n), dtype=np.int64)
        a = np.ascontiguousarray(am, dtype=np.int64)
        lib.aki_oracle_mma_mask(a.ctypes.data_as(C.c_void_p), C.c_int64(n), C.c_int64(s), C.c_int64(t), C.c_int64(e),
                                out.ctypes.data_as(C.c_void_p))
        assert np.array_equal(out, gen.unpack_mask_bits(g[f"bits_{i}"], (1, n, n))[0]), i
    ge = load_golden("tiny_e2e.npz")
    T = gen.TINY
    lib.aki_oracle_splice_src.restype = C.c_int64
    for b in range(ge["lang_x"].shape[0]):
        ids = np.ascontiguousarray(ge["lang_x"][b])
        kind = np.zeros(64, dtype=np.int32)
        idx = np.zeros(64, dtype=np.int64)
        L = lib.aki_oracle_splice_src(ids.ctypes.data_as(C.c_void_p), C.c_int64(len(ids)), C.c_int64(T["media_token_id"]),
                                      C.c_int64(T["num_vision_tokens"]), kind.ctypes.data_as(C.c_void_p), idx.ctypes.data_as(C.c_void_p))
        lab = np.where(kind[:L] == 1, -100, ge["labels"][b][np.clip(idx[:L], 0, len(ids) - 1)])
        assert np.array_equal(lab, ge["new_labels"][b][:L])
