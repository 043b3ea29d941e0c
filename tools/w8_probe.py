"""Upper bound for the 2-K-step form of the scan (tools; not part of the product): the BASELINE configs[1] batch against a
bank of 380 templates that are ALL 8x15 (the 95 8-wide templates four times) — 24 N-tiles at 2 K-steps, the MFMA work a
9-wide class would have with its ninth column bounded instead of multiplied."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from font_ocr_amd import Bank, synth_pages
from font_ocr_amd.searcher import Scanner, SCAN_MFMA
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
bank = Bank.load(os.path.join(ROOT, "tests/golden/bank_dejavu13_ascii95_x2.bin"))
pages = synth_pages(bank, 128, 608, 720)
sc = Scanner(0); sc.set_pages(pages)
eight = [t for t in range(len(bank)) if int(bank.templates[t]["n_w"]) == 8]
for name, sub in (("full bank (8x15 + 9x15)", bank), ("380 x 8x15", bank.subset(eight * 4)), ("95 x 8x15", bank.subset(eight))):
    sc.set_bank(sub)
    for _ in range(2): sc.scan(0.8, 1024, SCAN_MFMA)
    ms = []
    for _ in range(6):
        sc.scan(0.8, 1024, SCAN_MFMA)
        ms.append(sum(li["ms"] for li in sc.launches()))
    print(f"{name:28s} scan {np.mean(ms):.3f} ms (min {np.min(ms):.3f})  cand {sc.counters()['candidates']}  phases {sc.timings()}")
