"""Manual robustness check: threshold -1 on a multi-page batch (every finite window passes) -> candidate overflow ->
split-batch fallback.  Compares page 0 / last page with the oracle (tools only)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from font_ocr_amd import Bank, synth_pages
from font_ocr_amd.searcher import Scanner, SCAN_MFMA
from oracle import oracle as O
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
bank = Bank.load(os.path.join(ROOT, "tests/golden/bank_dejavu13_ascii95_x2.bin"))
P = int(os.environ.get("LT_PAGES", "16"))
pages = synth_pages(bank, P, 608, 720)
sc = Scanner(0); sc.set_bank(bank); sc.set_pages(pages)
t = time.time(); sc.scan(-1.0, 1024, SCAN_MFMA); dt = time.time() - t
counts = sc.counts(); c = sc.counters()
print(f"scan thr=-1 on {P} pages: {dt:.2f} s, candidates {c['candidates']:.3e}, raw hits {c['raw_hits']:.3e}, matches {sc.total_matches()}")
offsets, m = sc.matches()
T = len(bank)
for p in (0, P - 1):
    wc, wm = O.scan_page(O.invert(pages[p]), bank, -1.0, 1024, use_ref=O.have_ref())
    assert np.array_equal(counts[p], wc), p
    flat = np.concatenate([wm[t, : wc[t]] for t in range(T)])
    got = m[int(offsets[p * T]): int(offsets[(p + 1) * T])]
    assert got.tobytes() == flat.tobytes(), p
sc.process_hits(0.95, 5)
print("ok: identical to the reference lists on the checked pages;", sc.total_chars(), "chars after process_hits")
