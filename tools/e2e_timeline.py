"""Merged kernel + memory-copy timeline of a traced run (rocprofv3 --kernel-trace --memory-copy-trace): scan period and gaps,
DMA durations and where they lie, and a raw listing of a few periods in the steady state (one line per event, per queue).

    python tools/e2e_timeline.py <kernel_trace.csv> [<memory_copy_trace.csv>] [--list N]
"""
import csv, sys, collections
args = [x for x in sys.argv[1:] if not x.startswith("--")]
n_list = int(sys.argv[sys.argv.index("--list") + 1]) if "--list" in sys.argv else 4
rows = list(csv.DictReader(open(args[0])))
name_k = "Kernel_Name" if "Kernel_Name" in rows[0] else "Kernel Name"
q_k = "Stream_Id" if "Stream_Id" in rows[0] else "Queue_Id"
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r[name_k].split("(")[0].replace("void ", "").replace("focr::", "")[:44], "q" + r.get(q_k, "?")) for r in rows]
if len(args) > 1:
    for r in csv.DictReader(open(args[1])):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "DMA " + r.get("Direction", r.get("Kind", "")), "copy"))
ev.sort()
scans = [e for e in ev if "scan_mfma2" in e[2]]
lo, hi = len(scans) // 3, len(scans) - len(scans) // 5
ss = scans[lo:hi]
n = len(ss) - 1
t0, t1 = ss[0][0], ss[-1][0]
print("scan launches analysed: %d; period %.1f us; scan duration %.1f us; gap %.1f us" % (n, (t1 - t0) / n / 1e3, sum(e[1] - e[0] for e in ss[:-1]) / n / 1e3,
      sum(ss[i + 1][0] - ss[i][1] for i in range(n)) / n / 1e3))
dma = [e for e in ev if e[3] == "copy" and t0 <= e[0] < t1]
by = collections.defaultdict(list)
for e in dma: by[e[2]].append((e[1] - e[0]) / 1e3)
for k, v in by.items():
    big = [x for x in v if x > 100]
    print("%-28s %5d copies in the window, %d above 100 us: avg %.1f us (min %.1f max %.1f)" % (k, len(v), len(big), sum(big) / max(1, len(big)), min(big or [0]), max(big or [0])))
busy = collections.defaultdict(lambda: [0, 0.0])
for s, e, nm, q in ev:
    if t0 <= s < t1 and q != "copy":
        busy[nm][0] += 1; busy[nm][1] += (e - s) / 1e3
print("%-46s %10s %10s %12s" % ("kernel", "calls/scan", "avg us", "us per scan"))
for nm, b in sorted(busy.items(), key=lambda kv: -kv[1][1])[:30]:
    print("%-46s %10.2f %10.1f %12.1f" % (nm, b[0] / n, b[1] / b[0], b[1] / n))
# raw listing: n_list periods from the middle
mid = ss[len(ss) // 2][0]
end = ss[min(len(ss) - 1, len(ss) // 2 + n_list)][0]
print("\nraw events, %d scan periods (us from the first scan's start; events above 15 us or DMA):" % n_list)
for s, e, nm, q in ev:
    if mid <= s < end and ((e - s) > 15000 or q == "copy" and (e - s) > 5000):
        print("  %9.1f .. %9.1f  (%7.1f)  %-6s %s" % ((s - mid) / 1e3, (e - mid) / 1e3, (e - s) / 1e3, q, nm))
