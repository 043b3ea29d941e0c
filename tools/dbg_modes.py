"""Debug helper (tools; not part of the product): scenarios on fresh contexts, progress flushed line by line."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from font_ocr_amd import Bank, synth_pages
from font_ocr_amd.searcher import Scanner, SCAN_MFMA, SCAN_DIRECT, PREFILTER_LEGACY, PREFILTER_ONE_STAGE
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
bank = Bank.load(os.path.join(ROOT, "tests/golden/bank_dejavu13_ascii95_x2.bin"))
def say(*a):
    print(*a, flush=True)
M1, M0 = (SCAN_MFMA, PREFILTER_ONE_STAGE), (SCAN_MFMA, PREFILTER_LEGACY)
pages8 = synth_pages(bank, 8, 608, 720)
scen = {
    "A: fresh, P8, legacy kernel + rows": [(pages8, M0, True)],
    "B: fresh, P8, mfma + rows": [(pages8, M1, True)],
    "C: fresh, P8, mfma + legacy tail, then rows": [(pages8, M1, False), (pages8, M1, True)],
    "D: fresh, P8, direct then mfma rows": [(pages8, SCAN_DIRECT, True), (pages8, M1, True)],
    "E: fresh, P3": [(pages8[:3], M1, True)],
    "F: fresh, P5": [(pages8[:5], M1, True)],
}
for name, steps in scen.items():
    say(name)
    sc = Scanner(0); sc.set_bank(bank)
    try:
        for pg, mode, rows in steps:
            sc.set_pages(pg); sc.set_row_tail(rows)
            sc.scan(0.8, 1024, mode)
            c = sc.counters()
            say("   ok", c["candidates"], c["raw_hits"], sc.total_matches())
    except Exception as e:
        say("   FAILED:", e)
    sc.close()
say("all done")
