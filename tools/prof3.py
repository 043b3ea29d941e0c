"""Per-phase wave cycles of scan_mfma3_kernel (experiment build: make -C font_ocr_amd/csrc hip EXTRA=-DFOCR_MFMA3_PROF)."""
import os, sys, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from font_ocr_amd import Bank, synth_pages, _native as N
from font_ocr_amd.searcher import Scanner, SCAN_MFMA
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
bank = Bank.load(os.path.join(ROOT, "tests/golden/bank_dejavu13_ascii95_x2.bin"))
pages = synth_pages(bank, 128, 608, 720)
sc = Scanner(0); sc.set_bank(bank); sc.set_pages(pages); sc.set_prefilter(2)
lib = C.CDLL(os.path.join(N.LIB_DIR, "libfocr_hip.so"))
out = (C.c_ulonglong * 8)()
for _ in range(2): sc.scan(0.8, 1024, SCAN_MFMA)
lib.focr_debug_prof(out, 1)
n = 5
for _ in range(n): sc.scan(0.8, 1024, SCAN_MFMA)
lib.focr_debug_prof(out, 1)
v = np.array(list(out)[:5], float) / n
names = ["prologue(issue loads)", "stage1(+wait A)", "mid", "stage2", "stage3"]
tot = v.sum()
for nm, x in zip(names, v): print(f"{nm:24s} {x/1e6:10.1f} Mcycles  {100*x/tot:5.1f} %")
waves = 256 * 16
vis, items = out[5] / n, out[6] / n
print("visits per item", vis / items, "items", items, "flagged block fraction", vis / items / (4 * 24))
print("per wave total cycles", tot / waves, "kernel ms", [li["ms"] for li in sc.launches()])
