"""aki_amd: MI355X-native (gfx950) implementation of AKI's modality-mutual-attention forward path.

Public surface mirrors the reference's ``open_flamingo`` package (codes/open_flamingo/__init__.py:1-2):
``create_model_and_transforms`` and the model classes.  All arithmetic on the path runs in
libaki_mi355x.so (hand-written HIP); see include/aki_mi355x.h for the C ABI.
"""
from ._lib import AkiError  # noqa: F401

_LAZY = {"AKI": ("aki", "AKI"), "create_model_and_transforms": ("factory", "create_model_and_transforms"),
         "build_aki": ("factory", "build_aki"), "AKIPretrained": ("modeling_aki", "AKI")}


def __getattr__(name):  # lazy: `import aki_amd` alone must not pull torch (tooling, build script)
    if name in _LAZY:
        import importlib
        mod, attr = _LAZY[name]
        return getattr(importlib.import_module(f".{mod}", __name__), attr)
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")
