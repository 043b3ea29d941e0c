import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from font_ocr_amd import Bank, synth_pages
from font_ocr_amd.searcher import Scanner, SCAN_MFMA
bank = Bank.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests/golden/bank_dejavu13_ascii95_x2.bin"))
pages = synth_pages(bank, 128, 608, 720)
sc = Scanner(0); sc.set_pages(pages); sc.set_bank(bank); sc.set_prefilter(1)
for thr in (0.8, 0.9, 0.97, 0.995, 1.5):
    for _ in range(2): sc.scan(thr, 1024, SCAN_MFMA)
    ms = []
    for _ in range(5):
        sc.scan(thr, 1024, SCAN_MFMA)
        ms.append(sum(li["ms"] for li in sc.launches()))
    print(f"thr {thr}: scan {np.mean(ms):.3f} ms cand {sc.counters()['candidates']}")
