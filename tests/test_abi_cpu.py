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
    assert lib.aki_abi_version() == 7
    assert b"aligned" in lib.aki_strerror(-3)


def test_host_side_validation_without_gpu(lib):
    from aki_amd import _lib as L
    a = L.LinearArgs()                                            # all-null arguments
    assert lib.aki_linear_fwd(C.byref(a), None) == -1             # AKI_ERR_INVALID_ARG, nothing launched
    c = L.MmaAttnCoreArgs()
    assert lib.aki_mma_attn_core_fwd(C.byref(c), None, 0, None) == -1
    assert lib.aki_mma_attn_workspace_bytes(8, 32, 655, 96, 0) >= 3 * 8 * 32 * 655 * 96 * 2
    assert lib.aki_patch_embed_workspace_bytes(8, 336, 14, 0) >= 8 * 576 * 640 * 2


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
