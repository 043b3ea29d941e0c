set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "c2_page0 or small_batch or full_size_c2 or fuzz or soak or estimates or ragged or random_banks or c3_geometry or cap or process_hits or extreme" > gpurun_out/r3_f_tests.log 2>&1; rc=$?
tail -5 gpurun_out/r3_f_tests.log
[ $rc -eq 0 ] || exit $rc
python tools/kbench.py 2>&1 | tail -1 | tee gpurun_out/r3_f_kbench.log
bash tools/kprof.sh r3_f_rows
python bench.py --steps 100 --no-cpu-baseline > gpurun_out/r3_f_bench_rows.json 2> gpurun_out/r3_f_bench_rows.err || exit 1
python - <<'PY'
import json
for f in ("rows",):
    d = json.load(open(f"gpurun_out/r3_f_bench_{f}.json")); r = d["roofline"]
    print(f, d["value"], d["ms_per_step"], r["avg_kernel_ms"], r["frac"], r.get("frac_whole_step"), r.get("isolated_avg_kernel_ms"), d["phases_ms_per_step"])
PY
