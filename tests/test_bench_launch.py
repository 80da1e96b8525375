"""bench.py's launcher contract on a box WITHOUT the GPUs it is asked for: it must refuse, never run fewer ranks."""
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, drop=("WORLD_SIZE", "RANK", "LOCAL_RANK")):
    env = {k: v for k, v in os.environ.items() if k not in drop}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=300)


@pytest.mark.skipif(torch.cuda.is_available() and torch.cuda.device_count() >= 2, reason="box has the GPUs: the refusal path is not reachable")
def test_gpus_2_without_two_gpus_refuses_instead_of_running_one_rank():
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"])
    assert r.returncode != 0
    assert "needs 2 GPUs" in (r.stderr + r.stdout)
    assert '"n_gpus"' not in r.stdout          # no result line was printed


def test_world_size_must_equal_gpus():
    r = _run(["--gpus", "4", "--steps", "1", "--warmup", "0"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"}, drop=())
    assert r.returncode != 0
    assert "must equal --gpus" in (r.stderr + r.stdout)
    assert '"n_gpus"' not in r.stdout
