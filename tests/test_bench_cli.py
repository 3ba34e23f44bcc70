"""bench.py command-line contract that can be checked without a GPU: `--gpus N` must never time a smaller
world under the label n_gpus = N."""
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=300)


def test_gpus_2_refuses_when_fewer_gpus_are_visible():
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("this box has two GPUs")
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"])
    assert r.returncode != 0
    assert "--gpus 2" in r.stderr and "GPU(s) visible" in r.stderr
    assert r.stdout.strip() == ""             # no JSON line for a run that did not happen


def test_gpus_flag_must_agree_with_world_size():
    r = _run(["--gpus", "1", "--steps", "1", "--warmup", "0"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "contradicts WORLD_SIZE=2" in r.stderr
    assert r.stdout.strip() == ""
