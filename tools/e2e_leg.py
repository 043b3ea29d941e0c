"""Only the PCIe-inclusive pipelined leg of bench.py (every batch starts in page-locked host memory, one DMA per batch under the
other lanes' scans), or the resident leg with the same loop, for tracing (rocprofv3 --kernel-trace --memory-copy-trace -- python3
tools/e2e_leg.py ...).  Prints one JSON line: Gpx/s of the leg.

    python tools/e2e_leg.py [--resident] [--in-flight 3] [--steps 60] [--pins K]   (K page-locked source buffers, default = lanes)
"""
import argparse
import json
import os
import sys
import time
from collections import deque

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from font_ocr_amd import Bank, synth_pages  # noqa: E402
from font_ocr_amd.searcher import SCAN_MFMA, PinnedPages, Pipeline  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--resident", action="store_true")
ap.add_argument("--in-flight", type=int, default=3)
ap.add_argument("--steps", type=int, default=60)
ap.add_argument("--pages", type=int, default=128)
ap.add_argument("--pins", type=int, default=0)
ap.add_argument("--no-prefetch", action="store_true", help="round 3's form: every lane uploads its batch at the head of its own chain")
a = ap.parse_args()
R_W, R_H, P = 608, 720, a.pages
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
bank = Bank.load(os.path.join(root, "tests", "golden", "bank_dejavu13_ascii95_x2.bin"))
pipe = Pipeline(0, a.in_flight)
pipe.set_bank(bank)
N_OUT = len(pipe.scanners)  # batches outstanding at the host: lanes x contexts per lane
pages = synth_pages(bank, P, R_W, R_H)
n_pins = a.pins or N_OUT
pins = []
for j in range(n_pins):
    pin = PinnedPages(P, R_H, R_W)
    pin.array[:] = pages
    pins.append(pin)


def run(n, resident):
    tickets = deque()
    ahead = 0 if (resident or a.no_prefetch) else min(N_OUT, n)  # batches announced ahead of their submit (focr_pipe_prefetch)
    for k in range(ahead):
        pipe.prefetch(pins[k % n_pins].array)
    for k in range(n):
        if len(tickets) == N_OUT:
            t = tickets.popleft()
            pipe.wait(t)
            pipe.release(t)
        if k + 1 == n:
            pipe.announce_last()
        tickets.append(pipe.submit(None if resident else pins[k % n_pins].array, 0.8, 1024, SCAN_MFMA, True, 0.95, 5))
        if ahead and k + ahead < n:
            pipe.prefetch(pins[(k + ahead) % n_pins].array)
    while tickets:
        t = tickets.popleft()
        pipe.wait(t)
        pipe.release(t)


run(2 * N_OUT, False)  # every context holds pages, sizes are known
run(4 * N_OUT, a.resident)
t0 = time.perf_counter()
run(a.steps, a.resident)
dt = time.perf_counter() - t0
print(json.dumps({"leg": "resident" if a.resident else "h2d_pipelined", "in_flight": a.in_flight, "steps": a.steps,
                  "ms_per_step": round(dt / a.steps * 1e3, 4), "gpx_s": round(P * R_W * R_H * a.steps / dt / 1e9, 3)}))
pipe.close()
