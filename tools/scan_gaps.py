"""Gaps between consecutive scan kernels in a rocprofv3 kernel trace (tools; not part of the product).
usage: scan_gaps.py <kernel_trace.csv>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
name_key = "Kernel_Name" if "Kernel_Name" in rows[0] else "Name"
ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r[name_key]) for r in rows), key=lambda t: t[0])
scans = [k for k in ks if "scan_mfma" in k[2]]
t0 = scans[len(scans) // 4][0]; t1 = scans[-len(scans) // 4][1]
mid = [k for k in scans if k[0] >= t0 and k[1] <= t1]
dur = [e - s for s, e, _ in mid]
gaps = [mid[i + 1][0] - mid[i][1] for i in range(len(mid) - 1)]
print("scan kernels", len(mid), "avg dur us", sum(dur) / len(dur) / 1e3, "avg gap us", sum(gaps) / len(gaps) / 1e3, "min/max gap", min(gaps) / 1e3, max(gaps) / 1e3)
print("period us", (mid[-1][0] - mid[0][0]) / (len(mid) - 1) / 1e3)
# busy time of the other kernels inside the window, and how much of it overlaps a scan
oth = [k for k in ks if "scan_mfma" not in k[2] and k[0] >= t0 and k[1] <= t1]
print("other kernels", len(oth), "sum dur ms", sum(e - s for s, e, _ in oth) / 1e6, "window ms", (t1 - t0) / 1e6)
by = {}
for s, e, n in oth: by[n[:50]] = by.get(n[:50], 0) + (e - s)
for n, v in sorted(by.items(), key=lambda t: -t[1])[:14]: print("   %-52s %8.1f us per scan" % (n, v / 1e3 / len(mid)))
