"""bench.py --gpus N launches its N ranks itself: the launcher, rendezvous, barrier-bracketed timing,
MAX over ranks and the rank census, exercised on the CPU with the gloo backend and a stub step (--dry-launch); and the
refusal to measure fewer GPUs than asked for."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*args, env=None):
    e = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=300, env=e)


def test_gpus_2_dry_launch_starts_two_ranks():
    r = _run("--gpus", "2", "--dry-launch", "--steps", "4", "--warmup", "1")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout  # exactly one JSON line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and len(d["per_rank_value"]) == 2 and d["dry_launch"] is True
    assert d["steps"] == 4 and d["warmup"] == 1 and d["ms_per_step"] >= 2.0  # the slowest rank (2 ms per stub step) sets the job's time
    assert d["per_rank_value"][0] > d["per_rank_value"][1]  # rank 1's stub step is twice as long


def test_world_size_mismatch_is_refused():
    r = _run("--gpus", "2", "--dry-launch", "--steps", "1", "--warmup", "0", env={"RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE" in r.stderr


def test_more_gpus_than_present_is_refused_loudly():
    import torch

    if torch.cuda.device_count() >= 2:
        import pytest

        pytest.skip("needs a machine with fewer than 2 GPUs")
    r = _run("--gpus", "2", "--steps", "1", "--warmup", "0")
    assert r.returncode != 0 and "refusing" in r.stderr and r.stdout.strip() == ""
