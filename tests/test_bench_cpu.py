"""bench.py's launch logic, as far as it can be exercised without a GPU: `--gpus N` outside a launcher starts N ranks itself or
refuses - it never degrades to a one-rank run that would record a flat scaling line (missing #1)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, **env_over):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(env_over)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, env=env, timeout=600, cwd=ROOT)


def _json_lines(r):
    return [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_gpus_n_refuses_when_fewer_gpus_are_visible():
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("this box has the GPUs")
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"])
    assert r.returncode != 0 and not _json_lines(r)
    assert "only" in r.stderr and "GPUs are visible" in r.stderr


def test_gpus_n_spawns_ranks_and_relays_their_exit_code():
    """With the gloo hook the device-count gate is off, so the child launcher really starts two ranks; without a GPU each rank
    stops at the product path's 'needs an MI355X' check - the parent must come back non-zero, with no JSON line."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("covered on the GPU by tests/test_dp_gpu.py::test_bench_gpus_2_starts_its_own_ranks")
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"], AKI_BENCH_BACKEND="gloo")
    assert r.returncode != 0 and not _json_lines(r)
    assert "needs an MI355X" in (r.stderr + r.stdout)


def test_world_size_mismatch_is_an_error():
    r = _run(["--gpus", "4", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"], WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr
