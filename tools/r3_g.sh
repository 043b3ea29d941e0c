set -o pipefail
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r3_g_tests.log 2>&1; rc=$?
tail -5 gpurun_out/r3_g_tests.log
[ $rc -eq 0 ] || exit $rc
bash tools/kprof.sh r3_g_rows
python bench.py --steps 100 > gpurun_out/r3_g_bench.json 2> gpurun_out/r3_g_bench.err || { tail -5 gpurun_out/r3_g_bench.err; exit 1; }
for cus in 176 208 224 240 256; do python bench.py --steps 100 --no-cpu-baseline --no-e2e --scan-cus $cus > gpurun_out/r3_g_bench_cu$cus.json 2> /dev/null || exit 1; done
python bench.py --steps 100 --no-cpu-baseline --no-e2e --in-flight 2 --scan-cus 224 > gpurun_out/r3_g_bench_if2_cu224.json 2> /dev/null || exit 1
python bench.py --steps 100 --no-cpu-baseline --no-e2e --in-flight 2 --scan-cus 256 > gpurun_out/r3_g_bench_if2_cu256.json 2> /dev/null || exit 1
python bench.py --steps 100 --no-cpu-baseline --no-e2e --in-flight 4 > gpurun_out/r3_g_bench_if4.json 2> /dev/null || exit 1
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r3_g_bench*.json")):
    d = json.load(open(f)); r = d["roofline"]
    print(f.split("r3_g_")[1], d["value"], d["ms_per_step"], r["avg_kernel_ms"], r["frac"], r.get("frac_whole_step"), r.get("isolated_avg_kernel_ms"), d.get("e2e_value_incl_h2d_pipelined"), d["phases_ms_per_step"])
PY
