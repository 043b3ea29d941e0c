# round 3, GPU call C: quick parity subset, kbench (rows vs legacy tail), bench
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "c2_page0 or small_batch or full_size_c2 or fuzz or soak or estimates or ragged or random_banks" > gpurun_out/r3_c_tests.log 2>&1; rc=$?
tail -5 gpurun_out/r3_c_tests.log
[ $rc -eq 0 ] || exit $rc
python tools/kbench.py 2>&1 | tail -1 | tee gpurun_out/r3_c_kbench.log
KB_LEGACY_TAIL=1 python tools/kbench.py 2>&1 | tail -1 | tee -a gpurun_out/r3_c_kbench.log
python bench.py --steps 100 --no-cpu-baseline > gpurun_out/r3_c_bench_rows.json 2> gpurun_out/r3_c_bench_rows.err || exit 1
python bench.py --steps 100 --no-cpu-baseline --legacy-tail > gpurun_out/r3_c_bench_legacy.json 2> /dev/null || exit 1
python - <<'PY'
import json
for f in ("rows", "legacy"):
    d = json.load(open(f"gpurun_out/r3_c_bench_{f}.json")); r = d["roofline"]
    print(f, d["value"], d["ms_per_step"], r["avg_kernel_ms"], r["frac"], r.get("frac_whole_step"), r.get("isolated_avg_kernel_ms"), d["phases_ms_per_step"])
PY
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r3_c_prof -o kb -- python3 $GRAFT_REPO_ROOT/tools/kbench.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT && python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/r3_c_prof/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
calls = max(int(r["Calls"]) for r in rows if "scan_mfma2s" in r["Name"])
for r in rows[:28]:
    n = r["Name"]; short = n.split("(")[0][-60:]
    print(f"{short:62s} calls/scan={int(r['Calls'])/calls:6.2f} avg_us={float(r['AverageNs'])/1e3:9.1f} per_scan_us={float(r['TotalDurationNs'])/calls/1e3:9.1f}")
PY
