"""round 5: the lanes' rhythm from a bench run's per-ticket dump (FOCR_BENCH_DUMP_TICKETS=<file> python bench.py ...): for a window of
tickets, where each batch's phases lie on the device's clock (focr_debug_phase_stamps) — the gaps between consecutive scan launches,
how long a batch waited between the end of its statistics and the start of its scan, how long its tail took."""
import json, sys
t = json.load(open(sys.argv[1]))
lo = int(sys.argv[2]) if len(sys.argv) > 2 else len(t) // 2
hi = lo + (int(sys.argv[3]) if len(sys.argv) > 3 else 18)
t0 = t[lo]["scan_launch_start"]
print("ticket lane  stats[start..end]   scan[start..end]  gap-to-prev-scan  wait-for-turn   tail-end  post-end   (ms, relative)")
prev_end = None
for x in t[lo:hi]:
    r = lambda k: x[k] - t0
    gap = (x["scan_launch_start"] - prev_end) if prev_end is not None else float("nan")
    print("%6d %4d  %7.2f .. %7.2f  %7.2f .. %7.2f  %8.3f  %12.3f  %9.2f %9.2f" % (x["ticket"], (x["ticket"] - 1) % 3, r("stats_start"), r("stats_end"), r("scan_launch_start"), r("scan_launch_end"),
          gap, x["scan_launch_start"] - x["stats_end"], r("order_end"), r("post_end")))
    prev_end = x["scan_launch_end"]
import statistics
gaps = [b["scan_launch_start"] - a["scan_launch_end"] for a, b in zip(t, t[1:])]
dur = [x["scan_launch_end"] - x["scan_launch_start"] for x in t]
print("all tickets: scan duration mean %.3f, gap mean %.3f (p50 %.3f, max %.3f); stats duration mean %.3f; tail (scan end -> post end) mean %.3f" % (
    statistics.mean(dur), statistics.mean(gaps), statistics.median(gaps), max(gaps), statistics.mean(x["stats_end"] - x["stats_start"] for x in t),
    statistics.mean(x["post_end"] - x["scan_launch_end"] for x in t)))
