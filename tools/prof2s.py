"""Per-phase wave time of scan_mfma2s_kernel (experiment build: make -C font_ocr_amd/csrc hip EXTRA=-DFOCR_V2S_PROF)."""
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
out = (C.c_ulonglong * 8)()
thr = float(os.environ.get("KB_THR", "0.8"))
for _ in range(2): sc.scan(thr, 1024, SCAN_MFMA)
lib.focr_debug_prof2(out, 1)
n = 5
ms = 0.0
for _ in range(n):
    sc.scan(thr, 1024, SCAN_MFMA)
    ms += sum(li["ms"] for li in sc.launches() if "scan_mfma" in li["name"]) / n
lib.focr_debug_prof2(out, 1)
v = np.array(list(out), float) / n
names = ["item setup + load issue", "wait for the item's loads", "N-tile loop (MFMA)", "candidate visits", "flushes", "ticket wait"]
tot = v[:6].sum()
visits, flushes, items = int(out[6]) & 0xffffffff, int(out[6]) >> 32, out[7]
waves = 256 * 16
print(f"thr {thr} kernel {ms:.3f} ms; wave time total {tot / waves:.0f} ticks per wave -> {tot / waves / ms / 1e3:.1f} ticks/us")
for nm, x in zip(names, v[:6]): print(f"  {nm:28s} {100 * x / tot:5.1f} %   {x / waves / (tot / waves / ms):.4f} ms of each wave")
print(f"  items {items / n:.0f}  visits/item {visits / items:.2f}  flushes/item {flushes / items:.3f}  ticks per visit {v[3] * n / max(visits, 1):.0f}  ticks per flush {v[4] * n / max(flushes, 1):.0f}")
