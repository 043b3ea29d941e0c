"""Time-budgeted fuzz of the whole device path against the reference kernel (oracle/_ref) — the long form of
test_gpu_parity.py::test_fuzz_geometry_banks_thresholds.  In the suite it runs for FOCR_FUZZ_SECONDS (default 15 s); the committed
evidence (profiles/r04_fuzz_long.log) is a run of several minutes:  FOCR_FUZZ_SECONDS=600 pytest tests/test_gpu_fuzz_long.py -m gpu -s

Every iteration draws a geometry, a bank (1-3 size classes, all K layouts, sometimes large enough that the verify runs in chunk passes),
pages (noise / mostly paper / planted templates), a threshold (negative ones included) and a cap, and compares, list for list and bit
for bit (x, y, f32 similarity, order, cap):
  * the MFMA path under both tails, each scanned TWICE (the second scan of a setup runs on the first one's size estimates),
  * the exact v_dot4 path,
  * process_hits on the device against the oracle's on the same lists,
  * and, every fourth iteration, the executor: three lanes, batches announced ahead with focr_pipe_prefetch, results per ticket.
"""
import os
import time

import numpy as np
import pytest

from font_ocr_amd.searcher import SCAN_DIRECT, PinnedPages, Pipeline, Scanner
from oracle import oracle as O
from test_gpu_parity import MFMA1, _assert_same, _csr_to_lists, _oracle_lists, _random_bank, _scan

pytestmark = pytest.mark.gpu


def _draw(rng, it):
    n_classes = int(rng.integers(1, 4))
    kind = it % 6
    if kind == 0:  # font-like
        shapes = [(int(rng.integers(8, 13)), int(rng.choice([14, 15, 16]))) for _ in range(n_classes)]
        per_shape = int(rng.integers(4, 40))
    elif kind == 1:  # column-drop widths, alone and beside their kept box
        shapes = [(int(rng.choice([8, 9, 9, 12, 13, 13])), int(rng.choice([13, 15, 16]))) for _ in range(n_classes)]
        per_shape = int(rng.integers(40, 120))
    elif kind == 2:  # a bank above the LDS: the verify runs in chunk passes
        shapes = [(int(rng.integers(9, 17)), int(rng.integers(12, 25))) for _ in range(min(n_classes, 2))]
        per_shape = int(rng.integers(250, 420))
    else:
        shapes = [(int(rng.integers(1, 17)), int(rng.integers(1, 33))) for _ in range(n_classes)]
        per_shape = int(rng.integers(1, 24))
    bank = _random_bank(rng, shapes, per_shape)
    big_bank = len(bank) > 200
    n_pages = int(rng.integers(1, 3 if big_bank else 7))
    r_w = int(rng.integers(max(s[0] for s in shapes) + 1, 120 if big_bank else 330))
    r_h = int(rng.integers(max(s[1] for s in shapes) + 1, 70 if big_bank else 200))
    thr = float(rng.choice([-1.0, -0.5, 0.0, 0.1, 0.4, 0.8, 0.97]))
    cap = int(rng.choice([1, 2, 37, 1024]))
    return shapes, bank, n_pages, r_w, r_h, thr, cap


def _pages(rng, bank, n_pages, r_w, r_h, style):
    pages = rng.integers(0, 256, (n_pages, r_h, r_w), dtype=np.uint8)
    if style == 0:
        pages[rng.random(pages.shape) < 0.85] = 255  # mostly paper: the blank-tile skip
    elif style == 1:
        pages[:] = 255
        pages[:, :: max(2, r_h // 7)] = rng.integers(0, 256, pages[:, :: max(2, r_h // 7)].shape, dtype=np.uint8)  # ink only on a few rows
    for k in range(0, len(bank), max(1, len(bank) // 40)):  # plant templates: high similarities exist, ties between neighbours too
        nd = bank.needle(k)
        p, y, x = int(rng.integers(0, n_pages)), int(rng.integers(0, max(1, r_h - nd.shape[0]))), int(rng.integers(0, max(1, r_w - nd.shape[1])))
        h, w = min(nd.shape[0], r_h - y), min(nd.shape[1], r_w - x)
        pages[p, y:y + h, x:x + w] = 255 - nd[:h, :w]
    return pages


def test_fuzz_for_a_time_budget():
    budget = float(os.environ.get("FOCR_FUZZ_SECONDS", "15"))
    seed = int(os.environ.get("FOCR_FUZZ_SEED", "4"))
    t_end = time.monotonic() + budget
    sc = Scanner(0)
    stat = dict(iterations=0, scans=0, matches=0, chars=0, capped=0, chunked=0, estimated=0, pipeline_batches=0)
    it = 0
    t_note = time.monotonic() + 30
    try:
        while time.monotonic() < t_end:
            rng = np.random.default_rng([seed, it])
            shapes, bank, n_pages, r_w, r_h, thr, cap = _draw(rng, it)
            pages = _pages(rng, bank, n_pages, r_w, r_h, it % 3)
            want = _oracle_lists(pages, bank, thr, cap)
            what = f"seed {seed} it {it} shapes={shapes} x{len(bank)} {n_pages}p {r_w}x{r_h} thr={thr} cap={cap}"
            sc.set_column_drop(True)
            for tail in (1, 0):
                sc.set_row_tail(tail)
                sc.set_bank(bank)
                sc.set_pages(pages)
                for rep in range(2):  # the second scan of the same setup runs on the first one's size estimates
                    _scan(sc, bank, thr, cap, MFMA1)
                    offsets, m = sc.matches()
                    _assert_same(_csr_to_lists(offsets, m, n_pages, len(bank)), want, f"{what} tail={tail} rep={rep}")
                    stat["scans"] += 1
                    stat["estimated"] += rep
                stat["matches"] += len(m)
                stat["capped"] += int((sc.counts() == cap).sum())
            stat["chunked"] += int(len(bank) > 200)
            sc.set_row_tail(1)
            _scan(sc, bank, thr, cap, SCAN_DIRECT)
            offsets, m = sc.matches()
            _assert_same(_csr_to_lists(offsets, m, n_pages, len(bank)), want, f"{what} direct")
            anchor, overlap = float(rng.choice([0.3, 0.6, 0.95])), int(rng.integers(0, 8))
            sc.process_hits(anchor, overlap)
            lines = sc.lines()
            for p in range(n_pages):
                counts = np.array([len(x) for x in want[p]], np.uint32)
                mm = np.zeros((len(bank), max(cap, 1)), O.MATCH_DTYPE)
                for t, x in enumerate(want[p]):
                    mm[t, : len(x)] = x
                if counts.sum() == 0:  # the reference panics on an empty hit list (src/ncc.rs:1040); the device reports no lines
                    assert len(lines[p]) == 0, what
                    continue
                wl = O.process_hits(O.raw_hits(counts, mm, bank), anchor, overlap)
                assert len(lines[p]) == len(wl), (what, p)
                for lg, lw in zip(lines[p], wl):
                    assert np.array_equal(lg["x"].astype(np.int64), lw["x"].astype(np.int64)) and np.array_equal(lg["letter"], lw["letter"]), (what, p)
                    assert lg["similarity"].tobytes() == lw["similarity"].tobytes(), (what, p)
                    stat["chars"] += len(lg)
            if it % 4 == 3 and len(bank) <= 200:  # the executor: three lanes, announced batches, results per ticket
                n_batches, n_lanes = 5, 3
                pins, wants = [], []
                for b in range(n_batches):
                    pin = PinnedPages(n_pages, r_h, r_w)
                    pin.array[:] = _pages(rng, bank, n_pages, r_w, r_h, (it + b) % 3)
                    pins.append(pin)
                    wants.append(_oracle_lists(pin.array, bank, thr, cap))
                pipe = Pipeline(0, n_lanes)
                try:
                    pipe.set_bank(bank)
                    for b in range(min(n_lanes, n_batches)):
                        pipe.prefetch(pins[b].array)
                    tickets = []
                    for b in range(n_batches + n_lanes):
                        if b >= n_lanes:
                            s2 = pipe.wait(tickets[b - n_lanes])
                            offsets, m = s2.matches()
                            _assert_same(_csr_to_lists(offsets, m, n_pages, len(bank)), wants[b - n_lanes], f"{what} pipeline batch {b - n_lanes}")
                            pipe.release(tickets[b - n_lanes])
                            stat["pipeline_batches"] += 1
                        if b == n_batches:
                            pipe.end_of_stream()
                        if b < n_batches:
                            tickets.append(pipe.submit(pins[b].array, thr, cap))
                            if b + n_lanes < n_batches:
                                pipe.prefetch(pins[b + n_lanes].array)
                finally:
                    pipe.close()
                    for pin in pins:
                        pin.close()
            it += 1
            stat["iterations"] = it
            if time.monotonic() > t_note:  # a long run shows that it is alive
                t_note = time.monotonic() + 30
                print(f"fuzz_long: {it} iterations, {stat['scans']} scans, {stat['matches']} matches so far", flush=True)
    finally:
        sc.close()
    line = f"fuzz_long: seed {seed}, {budget:.0f} s budget: " + ", ".join(f"{k} {v}" for k, v in stat.items()) + "; 0 mismatches"
    print(line)
    if os.path.isdir("gpurun_out"):
        with open(os.path.join("gpurun_out", "fuzz_long.log"), "a") as f:
            f.write(line + "\n")
    assert stat["iterations"] >= 1
