# CU occupancy of every kernel of one batch alone on the chip (tools/kbench.py): SQ_BUSY_CU_CYCLES per kernel, normalised by the scan kernel
# (one workgroup on every CU for its whole duration) -> CU-milliseconds per kernel: what a kernel costs the chip when batches share it.
# usage: bash tools/pmc_cu_busy.sh <tag> [KB_...=...]
tag=${1:-cu}; shift
repo=$PWD; out=$PWD/gpurun_out/pmc_$tag; mkdir -p $out
for kv in "$@"; do export "$kv"; done
(cd /tmp && TMPDIR=/tmp rocprofv3 --pmc SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/p -o p -- python3 $repo/tools/kbench.py > $out/run.log 2>&1) || echo "pmc pass failed"
f=$(find $out/p -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/${tag}_cu_busy.csv; rm -rf $out
python3 - $tag <<'PY'
import csv, sys, collections
by = collections.defaultdict(lambda: collections.defaultdict(list))
for x in csv.DictReader(open(f"gpurun_out/{sys.argv[1]}_cu_busy.csv")):
    n = x["Kernel_Name"].split("(")[0].replace("void ", "").replace("focr::", "")[:48]
    by[n][x["Counter_Name"]].append(float(x["Counter_Value"]))
    by[n]["dur_us"].append((int(x["End_Timestamp"]) - int(x["Start_Timestamp"])) / 1e3)
avg = {n: {k: sum(v) / len(v) for k, v in c.items()} for n, c in by.items()}
calls = {n: len(c["dur_us"]) / 3 for n, c in by.items()}  # three counters -> three rows per dispatch
scan = max((n for n in avg if "scan_mfma2" in n), key=lambda n: avg[n]["dur_us"])
n_scan = calls[scan]
unit = 256.0 * avg[scan]["dur_us"] / 1e3 / avg[scan]["SQ_BUSY_CU_CYCLES"]  # CU-ms per counted unit
print(f"{'kernel':50s} {'calls/batch':>11s} {'us alone':>9s} {'CU-ms/batch':>11s} {'avg CUs busy':>12s}")
tot = 0
for n, a in sorted(avg.items(), key=lambda kv: -kv[1]["SQ_BUSY_CU_CYCLES"] * calls[kv[0]]):
    cums = a["SQ_BUSY_CU_CYCLES"] * unit * calls[n] / n_scan
    tot += cums if n != scan else 0
    print(f"{n:50s} {calls[n] / n_scan:11.2f} {a['dur_us']:9.1f} {cums:11.2f} {a['SQ_BUSY_CU_CYCLES'] * unit / (a['dur_us'] / 1e3):12.1f}")
print("everything but the scan kernel: %.1f CU-ms per batch" % tot)
PY
