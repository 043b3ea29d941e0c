"""Soak test of the three-lane pipeline (tools; not part of the product): many different batches through the MFMA path with
estimated sizes and the scans taking turns, every batch's matches and lines compared with the direct (v_dot4, no prefilter,
no item queue) scan of the same pages on a fourth context.

    python tools/stress_pipeline.py [--batches 120] [--pages 24] [--w 608] [--h 240] [--bank x2|x2y2] [--prefetch]
--prefetch: every batch lies in page-locked memory and is announced three batches ahead (focr_pipe_prefetch: copy + ingest into the
lane's alternate page set on its copy stream; the geometry changes every eleventh batch, so the sets are re-allocated on the way).
"""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from font_ocr_amd import Bank, synth_pages
from font_ocr_amd.searcher import PinnedPages, Pipeline, Scanner, SCAN_DIRECT, SCAN_MFMA

ap = argparse.ArgumentParser()
ap.add_argument("--batches", type=int, default=120)
ap.add_argument("--pages", type=int, default=24)
ap.add_argument("--w", type=int, default=608)
ap.add_argument("--h", type=int, default=240)
ap.add_argument("--bank", choices=["x2", "x2y2"], default="x2", help="x2y2 = BASELINE configs[2]'s 1 520 templates: the verify runs in chunk passes")
ap.add_argument("--prefetch", action="store_true")
a = ap.parse_args()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
bank = Bank.load(os.path.join(ROOT, f"tests/golden/bank_dejavu13_ascii95_{a.bank}.bin"))
rng = np.random.default_rng(7)
pipe = Pipeline(0, 3); pipe.set_bank(bank)
ref = Scanner(0); ref.set_bank(bank)
t0 = time.time(); bad = 0; redone = 0
inflight = []

def check(ticket, pages, thr):
    global bad
    sc = pipe.wait(ticket)
    offs, m = sc.matches(); lines = sc.lines_flat().copy(); cnt = sc.counts().copy()
    pipe.release(ticket)
    ref.set_pages(pages, invert=True)
    ref.scan(thr, 1024, SCAN_DIRECT); ref.process_hits(0.95, 5)
    o2, m2 = ref.matches()
    if not (np.array_equal(offs, o2) and m.tobytes() == m2.tobytes() and np.array_equal(cnt, ref.counts()) and lines.tobytes() == ref.lines_flat().tobytes()):
        bad += 1
        print("MISMATCH in batch with thr", thr, "matches", len(m), len(m2), flush=True)

def make(b):
    n = int(rng.integers(1, a.pages + 1))
    pages = synth_pages(bank, n, a.w, a.h, first=int(rng.integers(0, 100000)))
    if b % 7 == 3: pages[rng.integers(0, n)] = 255  # a blank (white luma) page in the batch
    if b % 11 == 5: pages = pages[:, : a.h - int(rng.integers(1, 40)), : a.w - int(rng.integers(1, 90))].copy()  # another geometry
    thr = float(rng.choice([0.8, 0.8, 0.8, 0.6, 0.9]))
    pin = None
    if a.prefetch:
        pin = PinnedPages(*pages.shape)
        pin.array[:] = pages
        pages = pin.array
    return pages, thr, pin

announced = []
made = 0
for b in range(a.batches):
    while a.prefetch and made < a.batches and len(announced) < 3:
        announced.append(make(made)); made += 1
        pipe.prefetch(announced[-1][0])
    pages, thr, pin = announced.pop(0) if a.prefetch else make(b)
    t = pipe.submit(pages, thr, 1024, SCAN_MFMA, True, 0.95, 5)
    inflight.append((t, pages, thr, pin))
    if len(inflight) == 3:
        t_, pg_, thr_, pin_ = inflight.pop(0)
        check(t_, pg_, thr_)
        if pin_ is not None: pin_.close()
    if b % 20 == 19: print("batch", b + 1, "mismatches", bad, "elapsed %.1f s" % (time.time() - t0), flush=True)
while inflight:
    t_, pg_, thr_, pin_ = inflight.pop(0)
    check(t_, pg_, thr_)
    if pin_ is not None: pin_.close()
print("done:", a.batches, "batches,", bad, "mismatches, %.1f s" % (time.time() - t0))
sys.exit(1 if bad else 0)
