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


def _failing(stage):
    import time

    t0 = time.time()
    r = _run("--gpus", "2", "--dry-launch", "--steps", "400", "--warmup", "1", env={"FOCR_BENCH_FAIL_RANK": f"1:{stage}", "FOCR_BENCH_INIT_TIMEOUT": "600"})
    return r, time.time() - t0


def test_a_rank_dying_before_the_rendezvous_fails_the_run_fast():
    """rank 1 dies before init_process_group; rank 0 would wait in the rendezvous for its timeout (set to
    600 s here).  The launcher must return rank 1's exit code within seconds, with rank 1's message, and leave nobody behind."""
    r, dt = _failing("init")
    assert r.returncode == 17, (r.returncode, r.stderr[-2000:])
    assert dt < 20.0, dt
    assert "[rank 1] bench.py: injected failure in rank 1 at stage 'init'" in r.stderr
    assert "rank 1 exited with code 17" in r.stderr and r.stdout.strip() == ""


def test_a_rank_dying_mid_run_fails_the_run_fast():
    """... and the same in the middle of the timed steps: rank 0 then waits in the closing barrier for a dead peer."""
    r, dt = _failing("step")
    assert r.returncode == 17, (r.returncode, r.stderr[-2000:])
    assert dt < 30.0, dt
    assert "[rank 1] bench.py: injected failure in rank 1 at stage 'step'" in r.stderr and r.stdout.strip() == ""


def test_ranks_stderr_is_prefixed():
    r = _run("--gpus", "2", "--dry-launch", "--steps", "2", "--warmup", "0", env={"FOCR_BENCH_DRY_CHATTER": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    assert "[rank 0] dry-launch rank 0 of 2" in r.stderr and "[rank 1] dry-launch rank 1 of 2" in r.stderr


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
