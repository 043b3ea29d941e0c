set -o pipefail
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r3_i_tests.log 2>&1; rc=$?
tail -5 gpurun_out/r3_i_tests.log
[ $rc -eq 0 ] || exit $rc
bash tools/kprof.sh r3_i_rows
python bench.py --steps 100 --no-cpu-baseline --no-e2e --scan-cus 224 > gpurun_out/r3_i_bench_cu224.json 2> /dev/null || exit 1
python bench.py --steps 100 --no-cpu-baseline --no-e2e > gpurun_out/r3_i_bench.json 2> gpurun_out/r3_i_bench.err || { tail -5 gpurun_out/r3_i_bench.err; exit 1; }
python bench.py --config c3 --pages-per-gpu 64 --steps 12 --warmup 2 --no-cpu-baseline --no-e2e > gpurun_out/r3_i_bench_c3.json 2> /dev/null || exit 1
python bench.py --config c3 --pages-per-gpu 64 --steps 12 --warmup 2 --no-cpu-baseline --no-e2e --scan-cus 224 > gpurun_out/r3_i_bench_c3_cu224.json 2> /dev/null || exit 1
python bench.py --noise --steps 100 --no-cpu-baseline --no-e2e > gpurun_out/r3_i_bench_noise.json 2> /dev/null || exit 1
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r3_i_bench*.json")):
    d = json.load(open(f)); r = d["roofline"]
    print(f.split("r3_i_")[1], d["value"], d["ms_per_step"], r["avg_kernel_ms"], r["frac"], r.get("frac_whole_step"), r.get("isolated_avg_kernel_ms"), d["phases_ms_per_step"], d["kernels_ms_per_step"])
PY
