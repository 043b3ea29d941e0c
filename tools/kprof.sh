# per-kernel durations of tools/kbench.py (one context, BASELINE configs[1], exact sizes) under rocprofv3 (run on the GPU box
# from the repo root): prints the table and leaves the CSV in gpurun_out/<tag>_kernel_stats.csv.  usage: bash tools/kprof.sh <tag> [env...]
tag=${1:-kprof}; shift
repo=$PWD; out=$PWD/gpurun_out/prof_$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
env "$@" true
for kv in "$@"; do export "$kv"; done
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o k -- python3 $repo/tools/kbench.py > $out/run.log 2>&1
cd $repo
f=$(find $out -name "*kernel_stats.csv" | head -1)
cp $f gpurun_out/${tag}_kernel_stats.csv
rm -rf $out
python3 - $tag <<'PY'
import csv, sys
rows = list(csv.DictReader(open(f"gpurun_out/{sys.argv[1]}_kernel_stats.csv")))
calls = max(int(r["Calls"]) for r in rows if "scan_mfma2" in r["Name"])
tot = 0
for r in rows:
    n = r["Name"]; short = n.split("(")[0][-64:]
    if "rocprim" in n: short = "rocprim:" + ("onesweep" if "onesweep" in n else "scan" if "scan" in n else "other") + (" pairs" if "float>" in n else "")
    per = float(r["TotalDurationNs"]) / calls / 1e3; tot += per
    if per > 2: print(f"{short:66s} calls/scan={int(r['Calls'])/calls:6.2f} avg_us={float(r['AverageNs'])/1e3:9.1f} per_scan_us={per:9.1f}")
print("sum per scan (us):", round(tot, 1))
PY
