"""BASELINE configs[4]: 256-template bank, windows x templates as an MFMA GEMM vs the v_dot4 ("LDS-NCC") kernel.
Prints per-kernel times for both device formulations on the same pages and checks that their results agree.
Run under rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE for the MFMA utilisation (tools only)."""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from font_ocr_amd import Bank, synth_pages
from font_ocr_amd.searcher import Scanner, SCAN_MFMA, SCAN_DIRECT
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
full = Bank.load(os.path.join(ROOT, "tests/golden/bank_dejavu13_ascii95_x2.bin"))
bank = full.subset(range(256))  # 95 glyphs at shift 0 + 161 variants at shifts 1/4, 1/2: 256 templates, two size classes
P = int(os.environ.get("C5_PAGES", "64"))
pages = synth_pages(full, P, 608, 720)
sc = Scanner(0); sc.set_bank(bank); sc.set_pages(pages)
res = {}
for name, mode in (("mfma_gemm", SCAN_MFMA), ("dot4_direct", SCAN_DIRECT)):
    for _ in range(2): sc.scan(0.8, 1024, mode)
    ms = {}
    for _ in range(5):
        sc.scan(0.8, 1024, mode)
        for li in sc.launches(): ms[li["name"]] = ms.get(li["name"], 0) + li["ms"] / 5
    alg = sum(li["alg_macs"] for li in sc.launches())
    res[name] = dict(kernels_ms={k: round(v, 3) for k, v in ms.items()}, scan_ms=round(sum(ms.values()), 3),
                     alg_TMACs=round(alg / sum(ms.values()) / 1e9, 1), counts_sum=int(sc.counts().sum()), matches=sc.matches()[1].tobytes())
assert res["mfma_gemm"]["matches"] == res["dot4_direct"]["matches"], "the two device formulations disagree"
for r in res.values(): r.pop("matches")
print(json.dumps(dict(workload=f"{P} pages 608x720, 256 templates", **res)))
