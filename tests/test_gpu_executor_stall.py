"""The executor's tolerance to a late host (round 5): a submitting thread that sleeps 3 ms every tenth step must not cost throughput —
the device works on the batches that are already queued (lanes x depth contexts, every batch queued when it is submitted)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*extra):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "150", "--warmup", "6", "--no-cpu-baseline", "--no-e2e", "--no-extra-legs", "--settle-s", "0.4", *extra],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


def test_injected_host_stalls_cost_under_three_per_cent():
    base = _bench()
    stalled = _bench("--inject-stall-ms", "3", "--inject-stall-every", "10")
    st = stalled["step_stats"]
    assert st["injected_host_stalls"] == 14 and st["longest_host_gap_ms"] >= 3.0, st
    # measured on the builder's boxes: 32.8 against 32.8 Gpx/s (0 %); one context per lane (--depth 1): -3 %; rounds 2-4's executor: -15 %
    assert stalled["value"] >= 0.97 * base["value"], (base["value"], stalled["value"], st)
    # and the line says where the time went (one box in five showed a single 6 ms device-side interval in one of the two runs with
    # the rate still inside the bound: the intervals are a diagnostic, not asserted on)
    assert st["device_interval_ms_p50"] is not None and st["device_interval_ms_max"] >= st["device_interval_ms_p50"], st
    assert base["parity"].startswith("page 0:") and "MISMATCH" not in base["parity"]
