"""Where the scan kernel's time goes: an experiment build (-DFOCR_V2S_VARIANTS, see scan_mfma2.hip) with parts of the item loop
switched off at run time — 1: no per-N-tile test (and nothing behind it), 2: test but no candidate visits, 4: visits stop at the
per-M-tile check, 8: no window / plane loads (synthetic operands), 16: synthetic live list.  Results are wrong by construction;
only the kernel's duration means anything.  FOCR_HIP_LIB=<variant build> python tools/variants.py"""
import os, sys
import ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from font_ocr_amd import Bank, synth_pages, _native
from font_ocr_amd.searcher import Scanner, SCAN_MFMA
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
bank = Bank.load(os.path.join(ROOT, "tests/golden/bank_dejavu13_ascii95_x2.bin"))
pages = synth_pages(bank, 128, 608, 720)
lib = _native.hip()
sc = Scanner(0)
sc.set_bank(bank); sc.set_pages(pages); sc.set_size_estimates(False)
for v in ([0] if os.environ.get("VARIANTS_COUNT_ONLY") else [0, 4, 2, 1, 10, 9, 0]):
    assert lib.focr_debug_v2s_variant(v) == 0
    for _ in range(2): sc.scan(0.8, 1024, SCAN_MFMA)
    acc = 0.0
    for _ in range(4):
        sc.scan(0.8, 1024, SCAN_MFMA)
        acc += sum(li["ms"] for li in sc.launches() if "scan_mfma2s" in li["name"]) / 4
    cnt = (C.c_ulonglong * 4)()
    lib.focr_debug_v2s_counts(cnt, 1)
    print("variant %2d: scan kernel %.3f ms, candidates %d; per scan: items %d, visits %d, M-tiles looked into %d, registers with a candidate %d"
          % (v, acc, sc.counters()["candidates"], cnt[0] // 6, cnt[1] // 6, cnt[2] // 6, cnt[3] // 6), flush=True)
