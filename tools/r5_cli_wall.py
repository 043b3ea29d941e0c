"""round 5: `ncc` wall time against its own 'total since main()' for several page counts (files from tools/cli_e2e.py), stdout to a pipe or to /dev/null"""
import os, subprocess, sys, time, re, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from font_ocr_amd import ASCII95
NCC = "font_ocr_amd/bin/ncc"; FONT = "/usr/share/fonts/truetype/dejavu/DejaVuSansMono.ttf"
paths = sorted(p for p in os.listdir("/tmp/focr_e2e") if p.startswith("p") and p.endswith(".pgm"))
paths = [os.path.join("/tmp/focr_e2e", p) for p in paths]
for n in (4096,):
    for sink in ("null", "file", "shm"):
        for extra in ({},):
            walls, mains = [], []
            for _ in range(5):
                env = dict(os.environ, FOCR_CLI_TIMING="1", **extra)
                t0 = time.perf_counter()
                out = {"null": subprocess.DEVNULL, "pipe": subprocess.PIPE, "file": open("/tmp/focr_e2e/out.txt", "wb"), "shm": open("/dev/shm/focr_out.txt", "wb")}[sink]
                r = subprocess.run([NCC, "-f", FONT, "-t", "13", "--x-bits", "2", "-a", ASCII95, "-i"] + paths[:n], env=env, stdout=out, stderr=subprocess.PIPE)
                walls.append((time.perf_counter() - t0) * 1e3)
                m = re.search(r"total since main\(\)\s+([\d.]+) ms", r.stderr.decode())
                mains.append(float(m.group(1)) if m else -1)
            print(f"pages {n:5d} stdout {sink:4s} teardown {'yes' if extra else 'no ':3s}: wall min {min(walls):7.1f} median {statistics.median(walls):7.1f} ms; since main() min {min(mains):7.1f} median {statistics.median(mains):7.1f} ms")
