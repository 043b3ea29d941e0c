"""Kernel-level timing of the scan at configs[1] (tools; not part of the product)."""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from font_ocr_amd import Bank, synth_pages
from font_ocr_amd.searcher import Scanner, SCAN_MFMA
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
C3 = os.environ.get("KB_CONFIG", "c2") == "c3"  # BASELINE configs[2] geometry: 1200x1600 pages, 1520 templates
bank = Bank.load(os.path.join(ROOT, "tests/golden/bank_dejavu13_ascii95_x2y2.bin" if C3 else "tests/golden/bank_dejavu13_ascii95_x2.bin"))
P = int(os.environ.get("KB_PAGES", "64" if C3 else "128"))
pages = synth_pages(bank, P, 1200 if C3 else 608, 1600 if C3 else 720)
sc = Scanner(0); sc.set_bank(bank); sc.set_pages(pages)
sc.set_prefilter(int(os.environ.get("KB_PREFILTER", "0")))
if os.environ.get("KB_LEGACY_TAIL"): sc.set_row_tail(0)
if os.environ.get("KB_TAIL"): sc.set_row_tail(int(os.environ["KB_TAIL"]))  # 1 = hits-first (default), 0 = legacy
if os.environ.get("KB_SCAN_CUS"): sc.set_scan_cus(int(os.environ["KB_SCAN_CUS"]))
for _ in range(2): sc.scan(0.8, 1024, SCAN_MFMA)
acc = {}
N = 5
POST = bool(os.environ.get("KB_POST"))  # also focr_process_hits(0.95, 5), as a bench step does
for _ in range(N):
    sc.scan(0.8, 1024, SCAN_MFMA)
    if POST: sc.process_hits(0.95, 5)
    for li in sc.launches(): acc[li["name"]] = acc.get(li["name"], 0) + li["ms"] / N
t = sc.timings(); c = sc.counters()
print("prefilter", os.environ.get("KB_PREFILTER", "0"), {k: round(v, 3) for k, v in acc.items()}, {k: round(v, 3) for k, v in t.items()}, c["candidates"], c["raw_hits"])
