"""Experiment: do two contexts driven by two host threads overlap on one GPU?  (tools only)

    python tools/two_ctx.py [--threads 2] [--steps 20] [--pages 128]

Every thread owns a context with its own resident C2 batch and runs scan + process_hits `steps` times; prints
the aggregate Gpx/s for 1..threads concurrent contexts."""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from font_ocr_amd import Bank, synth_pages  # noqa: E402
from font_ocr_amd.searcher import SCAN_MFMA, Scanner  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=2)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--pages", type=int, default=128)
    ap.add_argument("--cus", type=int, default=0, help="CUs for the scan kernel (0 = all)")
    ap.add_argument("--no-post", action="store_true", help="ablation: skip process_hits")
    a = ap.parse_args()
    bank = Bank.load(os.path.join(ROOT, "tests", "golden", "bank_dejavu13_ascii95_x2.bin"))
    pages = synth_pages(bank, a.pages, 608, 720)
    scs = []
    for _ in range(a.threads):
        sc = Scanner(0)
        sc.set_bank(bank)
        sc.set_scan_cus(a.cus)
        sc.set_pages(pages)
        for _ in range(2):
            sc.scan(0.8, 1024, SCAN_MFMA)
            sc.process_hits(0.95, 5)
        scs.append(sc)

    def work(sc):
        for _ in range(a.steps):
            sc.scan(0.8, 1024, SCAN_MFMA)
            if not a.no_post:
                sc.process_hits(0.95, 5)
        sc.sync()

    out = {}
    for n in range(1, a.threads + 1):
        ts = [threading.Thread(target=work, args=(scs[i],)) for i in range(n)]
        t0 = time.perf_counter()
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        dt = time.perf_counter() - t0
        out[f"{n}_ctx_Gpx_s"] = round(n * a.steps * a.pages * 608 * 720 / dt / 1e9, 2)
        out[f"{n}_ctx_ms_per_batch"] = round(dt / (n * a.steps) * 1e3, 3)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
