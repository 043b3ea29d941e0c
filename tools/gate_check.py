"""The executor queues its batches in ticket order (pipe.hip; rounds 3-4: a host-side gate in front of every scan launch).  Nothing
may hold the later tickets up: a batch that fails before its scan, a batch in direct mode (no turn in the scan kernels' chain), and
MFMA batches around them all complete, in order, with the right lists.  Run as a child process with a time limit by
tests/test_gpu_parity.py (a deadlock would otherwise hang the suite)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from font_ocr_amd import Bank, synth_pages
from font_ocr_amd.searcher import FocrError, Pipeline, SCAN_DIRECT, SCAN_MFMA

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
bank = Bank.load(os.path.join(ROOT, "tests/golden/bank_dejavu13_ascii95_x2.bin"))
pages = synth_pages(bank, 4, 320, 120)
pipe = Pipeline(0, 3)
pipe.set_bank(bank)
t1 = pipe.submit(pages[:2], 0.8, mode=SCAN_MFMA)
t2 = pipe.submit(None, 0.8, mode=SCAN_MFMA)      # context 2 holds no pages yet: fails before any kernel
t3 = pipe.submit(pages[2:], 0.8, mode=SCAN_DIRECT)  # no MFMA scan, so no turn to take
ref = {}
sc = pipe.wait(t1); ref[1] = sc.matches()[1].copy(); pipe.release(t1)
t4 = pipe.submit(pages[2:], 0.8, mode=SCAN_MFMA)    # (a submit blocks only while its own context's previous ticket is unreleased)
failed = False
try:
    pipe.wait(t2)
except FocrError:
    failed = True
pipe.release(t2)
sc = pipe.wait(t3); m3 = sc.matches()[1].copy(); pipe.release(t3)
sc = pipe.wait(t4); m4 = sc.matches()[1].copy(); pipe.release(t4)
assert failed, "the rescan of an empty lane should have failed"
assert len(m3) > 0 and np.array_equal(m3, m4), "direct and MFMA lists of the same pages differ"
for k in range(5, 11):  # and the lanes still roll
    t = pipe.submit(pages[:2], 0.8)
    assert t == k
    sc = pipe.wait(t)
    assert np.array_equal(sc.matches()[1], ref[1])
    pipe.release(t)
pipe.close()
print("gate ok")
