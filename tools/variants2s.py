"""Timing of scan_mfma2s_kernel with parts of the candidate path cut out (experiment build: make hip EXTRA=-DFOCR_V2S_VARIANTS)."""
import os, sys, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from font_ocr_amd import Bank, synth_pages, _native as N
from font_ocr_amd.searcher import Scanner, SCAN_MFMA
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
bank = Bank.load(os.path.join(ROOT, "tests/golden/bank_dejavu13_ascii95_x2.bin"))
pages = synth_pages(bank, 128, 608, 720)
sc = Scanner(0); sc.set_bank(bank); sc.set_pages(pages)
lib = C.CDLL(os.path.join(N.LIB_DIR, "libfocr_hip.so"))
for var in (0, 1, 9, 25, 8, 24, 0):
    lib.focr_debug_v2s_variant(var)
    for _ in range(2): sc.scan(0.8, 1024, SCAN_MFMA)
    ms = []
    for _ in range(8):
        sc.scan(0.8, 1024, SCAN_MFMA)
        ms.append(sum(li["ms"] for li in sc.launches() if "scan_mfma" in li["name"]))
    print("variant", var, "kernel ms", round(float(np.mean(ms)), 4), "min", round(min(ms), 4), "cand", sc.counters()["candidates"])
