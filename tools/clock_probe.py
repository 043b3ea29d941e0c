"""Sample the GPU's clocks and power (rocm-smi) while the scan runs back to back (tools; not part of the product).
KB_SCAN_CUS limits the CUs of the scan kernel."""
import os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from font_ocr_amd import Bank, synth_pages
from font_ocr_amd.searcher import Scanner, SCAN_MFMA
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
bank = Bank.load(os.path.join(ROOT, "tests/golden/bank_dejavu13_ascii95_x2.bin"))
pages = synth_pages(bank, 128, 608, 720)
sc = Scanner(0); sc.set_bank(bank); sc.set_pages(pages)
sc.set_size_estimates(True)
if os.environ.get("KB_SCAN_CUS"): sc.set_scan_cus(int(os.environ["KB_SCAN_CUS"]))
stop = False
samples = []
def sampler():
    while not stop:
        try:
            out = subprocess.run(["/opt/rocm/bin/rocm-smi", "-d", "0", "--showclocks", "--showpower", "--csv"], capture_output=True, text=True, timeout=10).stdout
            samples.append(out.strip().splitlines())
        except Exception as e:  # noqa
            samples.append([repr(e)])
        time.sleep(0.3)
for _ in range(3): sc.scan(0.8, 1024, SCAN_MFMA)
th = threading.Thread(target=sampler); th.start()
t0 = time.time(); n = 0; ms = 0.0
while time.time() - t0 < float(os.environ.get("KB_SECONDS", "6")):
    sc.scan(0.8, 1024, SCAN_MFMA); n += 1
    for li in sc.launches():
        if "scan_mfma" in li["name"]: ms += li["ms"]
stop = True; th.join()
print("scans", n, "avg scan kernel ms", round(ms / n, 4), "cus", os.environ.get("KB_SCAN_CUS", "all"))
for s in samples[:1] + samples[2::3]: print(s)
