"""CPU: bench.py's launch contract.  `--gpus N` must never be silently ignored: without a launcher it starts N ranks
itself or refuses; under a launcher the rank count must equal N."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(env_extra, *args):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)


def test_gpus_flag_without_enough_gpus_fails_loudly():
    import torch
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("this box could really launch two ranks")
    r = _run({}, "--gpus", "2", "--small")
    assert r.returncode != 0 and "refusing to report a 2-GPU figure" in r.stderr, (r.returncode, r.stderr[-500:])
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")], "a JSON line was printed for a job that did not run"


def test_gpus_flag_must_match_the_launcher_world_size():
    r = _run({"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"}, "--gpus", "2", "--small")
    assert r.returncode != 0 and "the two must agree" in r.stderr, (r.returncode, r.stderr[-500:])
