"""Timeline of a bench run from a rocprofv3 kernel trace (tools/timeline.sh): per scan launch its duration, the gap to the next
scan launch and which kernels ran in that gap; per kernel its average duration in flight and how much of it overlapped a scan."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
name_k = "Kernel_Name" if "Kernel_Name" in rows[0] else "Kernel Name"
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r[name_k].split("(")[0].replace("void ", "").replace("focr::", "")) for r in rows))
scans = [e for e in ev if "scan_mfma2" in e[2]]
# steady state: drop the first and last fifth
lo, hi = len(scans) // 5, len(scans) - len(scans) // 5
scans_s = scans[lo:hi]
t0, t1 = scans_s[0][0], scans_s[-1][0]
n = len(scans_s) - 1
print("scan launches analysed: %d; period %.1f us; scan duration %.1f us; gap scan-end -> next scan-start %.1f us" % (
    n, (t1 - t0) / n / 1e3, sum(e[1] - e[0] for e in scans_s[:-1]) / n / 1e3, sum(scans_s[i + 1][0] - scans_s[i][1] for i in range(n)) / n / 1e3))
# per kernel: time inside [t0, t1), overlapped with a scan or not
busy = collections.defaultdict(lambda: [0, 0.0, 0.0])
si = 0
for s, e, nm in ev:
    if e <= t0 or s >= t1: continue
    s, e = max(s, t0), min(e, t1)
    ov = 0
    for ss, se, _ in scans_s:
        if se <= s: continue
        if ss >= e: break
        ov += min(e, se) - max(s, ss)
    b = busy[nm]; b[0] += 1; b[1] += (e - s) / 1e3; b[2] += ov / 1e3
print("%-52s %8s %10s %12s %14s" % ("kernel", "calls/scan", "avg us", "us per scan", "of it under a scan"))
for nm, b in sorted(busy.items(), key=lambda kv: -kv[1][1]):
    print("%-52s %8.2f %10.1f %12.1f %14.1f" % (nm[:52], b[0] / n, b[1] / b[0], b[1] / n, b[2] / n))
# what fills the gaps between scans
gap = collections.defaultdict(float)
for i in range(n):
    gs, ge = scans_s[i][1], scans_s[i + 1][0]
    for s, e, nm in ev:
        if e <= gs or s >= ge or "scan_mfma2" in nm: continue
        gap[nm] += (min(e, ge) - max(s, gs)) / 1e3
print("kernels running while no scan runs (us per scan, summed over concurrent kernels):")
for nm, v in sorted(gap.items(), key=lambda kv: -kv[1])[:12]:
    print("   %-52s %8.1f" % (nm[:52], v / n))
