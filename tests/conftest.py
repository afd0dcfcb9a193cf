"""pytest configuration: ``gpu`` marker, import paths, golden loader."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TESTS = os.path.join(ROOT, "tests")
GOLDEN_DIR = os.path.join(TESTS, "golden")
for p in (ROOT, TESTS, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN_DIR, name), allow_pickle=False)


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(autouse=True)
def _poison_recycled_gpu_memory(request):
    """Before every GPU test: leave blocks full of 0xFF bytes (NaN as bf16 / f32, -1 as integers) in torch's caching
    allocator, so that `torch.empty` hands the test recycled memory that is NOT benign.  Kernels that touch unwritten memory
    "with probability 0" (clamped rows, padding columns) then fail loudly instead of passing by luck on zero-filled pages - the
    fused decode attention once did exactly that (0 * NaN from a stale KV-cache row)."""
    if "gpu" in request.keywords and _has_gpu():
        import torch
        junk = [torch.full((n,), 0xFF, dtype=torch.uint8, device="cuda") for n in (1 << 12, 1 << 16, 1 << 20, 1 << 24, 1 << 27, 3 << 27)]
        del junk
    yield
