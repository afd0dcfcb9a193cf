"""Import the sony/aki reference (read-only, /root/reference) in THIS container only.

Used exclusively by ``tests/golden/make_golden.py`` to produce the committed golden vectors.
Nothing here runs on the GPU box (``/root/reference`` does not exist there) and nothing from the
reference (source or bytecode) is copied into the repo: the fixtures hold inputs/expected outputs.

Three harness shims are needed to import the first-party reference files with the toolchain in this
image:

1. ``einops_exts`` is not installed -> a one-function ``rearrange_many`` stand-in in ``sys.modules``
   (reference use: ``src/helpers.py:9,91``).
2. ``src/vlm.py:9`` imports ``SiglipVisionTransformer`` from ``transformers.models.siglip.modeling_siglip``
   and ``src/vlm.py:202`` dispatches on the class *name*; transformers 5.x dropped that class name,
   so a subclass of ``SiglipVisionModel`` literally named ``SiglipVisionTransformer`` is injected.
3. The reference pins ``transformers==4.41.2`` (``codes/setup.py:9``) whose
   ``_prepare_4d_causal_attention_mask`` turns a 4-D 0/1 mask into an additive ``finfo.min`` mask.
   The installed 5.x passes 4-D masks through untouched, so :func:`wrap_lm_441_mask` applies the
   4.41.2 inversion in front of the language model.
"""
from __future__ import annotations

import importlib.util
import os
import sys
import types

import torch

REF_ROOT = "/root/reference/codes/open_flamingo"
_PKG = "aki_ref"


def _install_shims():
    if "einops_exts" not in sys.modules:
        from einops import rearrange

        m = types.ModuleType("einops_exts")

        def rearrange_many(tensors, pattern, **kw):
            return tuple(rearrange(t, pattern, **kw) for t in tensors)

        m.rearrange_many = rearrange_many
        sys.modules["einops_exts"] = m
    import transformers.models.siglip.modeling_siglip as ms

    if not hasattr(ms, "SiglipVisionTransformer"):
        class SiglipVisionTransformer(ms.SiglipVisionModel):  # noqa: D401 - name matters
            pass

        ms.SiglipVisionTransformer = SiglipVisionTransformer


def load_reference():
    """Returns a namespace with the reference modules utils/helpers/vlm/aki loaded by file path."""
    if not os.path.isdir(REF_ROOT):
        raise RuntimeError("reference tree not present; goldens can only be regenerated in the build container")
    _install_shims()
    if _PKG in sys.modules:
        return sys.modules[_PKG]
    pkg = types.ModuleType(_PKG)
    pkg.__path__ = [os.path.join(REF_ROOT, "src")]
    sys.modules[_PKG] = pkg
    sub = types.ModuleType(_PKG + ".src")
    sub.__path__ = [os.path.join(REF_ROOT, "src")]
    sys.modules[_PKG + ".src"] = sub
    for name in ("utils", "helpers", "vlm", "aki"):
        full = f"{_PKG}.src.{name}"
        spec = importlib.util.spec_from_file_location(full, os.path.join(REF_ROOT, "src", f"{name}.py"))
        mod = importlib.util.module_from_spec(spec)
        sys.modules[full] = mod
        spec.loader.exec_module(mod)
        setattr(sub, name, mod)
        setattr(pkg, name, mod)
    return pkg


def load_sft_loader_utils():
    """train/sft_data_utils/loader_utils.py (imports only its sibling templates/templates.py) loaded by file path."""
    root = os.path.join(REF_ROOT, "train", "sft_data_utils")
    if not os.path.isdir(root):
        raise RuntimeError("reference tree not present; goldens can only be regenerated in the build container")
    for name, path in (("aki_ref_sft", root), ("aki_ref_sft.templates", os.path.join(root, "templates"))):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__path__ = [path]
            sys.modules[name] = m
    mods = {}
    for full, path in (("aki_ref_sft.templates.templates", os.path.join(root, "templates", "templates.py")),
                       ("aki_ref_sft.loader_utils", os.path.join(root, "loader_utils.py"))):
        spec = importlib.util.spec_from_file_location(full, path)
        mod = importlib.util.module_from_spec(spec)
        sys.modules[full] = mod
        spec.loader.exec_module(mod)
        mods[full] = mod
    return mods["aki_ref_sft.loader_utils"]


def load_train_losses():
    """train/losses.py (imports torch only) loaded by file path."""
    path = os.path.join(REF_ROOT, "train", "losses.py")
    if not os.path.isfile(path):
        raise RuntimeError("reference tree not present; goldens can only be regenerated in the build container")
    spec = importlib.util.spec_from_file_location("aki_ref_train_losses", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def invert_mask_441(mask01: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """transformers==4.41.2 ``_prepare_4d_causal_attention_mask`` for a 4-D input:
    ``inverted = 1.0 - mask; inverted.masked_fill(inverted.bool(), finfo(dtype).min)``."""
    inverted = 1.0 - mask01.to(dtype)
    return inverted.masked_fill(inverted.to(torch.bool), torch.finfo(dtype).min)


def wrap_lm_441_mask(lang_model):
    """Give ``lang_model.forward`` the 4.41.2 mask semantics (shim 3)."""
    orig = lang_model.forward

    def fwd(*a, attention_mask=None, inputs_embeds=None, **kw):
        if attention_mask is not None and attention_mask.dim() == 4:
            dt = inputs_embeds.dtype if inputs_embeds is not None else torch.float32
            attention_mask = invert_mask_441(attention_mask, dt)
            if kw.get("position_ids") is None and inputs_embeds is not None:
                L = inputs_embeds.shape[1]
                kw["position_ids"] = torch.arange(L, device=inputs_embeds.device).unsqueeze(0)
        return orig(*a, attention_mask=attention_mask, inputs_embeds=inputs_embeds, **kw)

    lang_model.forward = fwd
    return lang_model
