set -o pipefail
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r3_h_tests.log 2>&1; rc=$?
tail -5 gpurun_out/r3_h_tests.log
[ $rc -eq 0 ] || exit $rc
bash tools/kprof.sh r3_h_rows
python bench.py --steps 100 --no-cpu-baseline --no-e2e > gpurun_out/r3_h_bench.json 2> gpurun_out/r3_h_bench.err || { tail -5 gpurun_out/r3_h_bench.err; exit 1; }
for cus in 208 224; do python bench.py --steps 100 --no-cpu-baseline --no-e2e --scan-cus $cus > gpurun_out/r3_h_bench_cu$cus.json 2> /dev/null || exit 1; done
python bench.py --steps 100 --no-cpu-baseline --no-e2e --in-flight 4 > gpurun_out/r3_h_bench_if4.json 2> /dev/null || exit 1
python bench.py --steps 100 --no-cpu-baseline --no-e2e --in-flight 4 --scan-cus 208 > gpurun_out/r3_h_bench_if4_cu208.json 2> /dev/null || exit 1
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r3_h_bench*.json")):
    d = json.load(open(f)); r = d["roofline"]
    print(f.split("r3_h_")[1], d["value"], d["ms_per_step"], r["avg_kernel_ms"], r["frac"], r.get("frac_whole_step"), r.get("isolated_avg_kernel_ms"), d["phases_ms_per_step"])
PY
