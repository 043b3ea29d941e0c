"""Scan-kernel time against the number of N-tiles (tools; not part of the product): how much of a launch is per-item
fixed cost (window-fragment loads, thresholds) and how much the MFMA loop."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from font_ocr_amd import Bank, synth_pages
from font_ocr_amd.searcher import Scanner, SCAN_MFMA
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
bank = Bank.load(os.path.join(ROOT, "tests/golden/bank_dejavu13_ascii95_x2.bin"))
pages = synth_pages(bank, 128, 608, 720)
sc = Scanner(0); sc.set_pages(pages)
sc.set_prefilter(int(os.environ.get("KB_PREFILTER", "1")))
nine = [t for t in range(len(bank)) if int(bank.templates[t]["n_w"]) == 9]
for n in (16, 48, 96, 144, 192, 240, 285):
    sub = bank.subset(nine[:n])
    sc.set_bank(sub)
    for _ in range(2): sc.scan(0.8, 1024, SCAN_MFMA)
    ms = []
    for _ in range(5):
        sc.scan(0.8, 1024, SCAN_MFMA)
        ms.append(sum(li["ms"] for li in sc.launches()))
    print(f"templates {n:4d} tiles {(n + 15) // 16:3d} scan {np.mean(ms):.3f} ms  cand {sc.counters()['candidates']}")
