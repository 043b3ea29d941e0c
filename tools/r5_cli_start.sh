#!/bin/bash
# round 5: where the `ncc` binary's start-up goes: process + dynamic loading (no HIP call), one page, and the loader's own statistics
N=font_ocr_amd/bin/ncc
F=/usr/share/fonts/truetype/dejavu/DejaVuSansMono.ttf
python - <<'PY'
import subprocess, time, os, sys
sys.path.insert(0, os.getcwd())
from font_ocr_amd import ASCII95, Bank, save_pgm, synth_page
from font_ocr_amd.bank import SYNTH_SEED_BASE
N="font_ocr_amd/bin/ncc"; F="/usr/share/fonts/truetype/dejavu/DejaVuSansMono.ttf"
os.makedirs("/tmp/focr_e2e", exist_ok=True)
bank = Bank.rasterize(F, 13, 2, 0, alphabet=ASCII95)
p="/tmp/focr_e2e/one.pgm"; save_pgm(p, synth_page(bank, SYNTH_SEED_BASE, 608, 720))
def t(cmd, env=None):
    best=1e9
    for _ in range(5):
        t0=time.perf_counter(); r=subprocess.run(cmd, capture_output=True, env=env); best=min(best,time.perf_counter()-t0)
    return best*1e3, r
print("true            %.1f ms" % t(["/bin/true"])[0])
print("ncc --help      %.1f ms" % t([N,"--help"])[0])
ms,r=t([N,"-f",F,"-t","13","--x-bits","2","-a",ASCII95,"-i",p], dict(os.environ, FOCR_CLI_TIMING="1"))
print("ncc one page    %.1f ms" % ms); print(r.stderr.decode()[-700:])
r=subprocess.run([N,"--help"], capture_output=True, env=dict(os.environ, LD_DEBUG="statistics"))
print(r.stderr.decode()[-900:])
PY
ldd $N | wc -l
