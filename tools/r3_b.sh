# round 3, GPU call B: parity suite with the row tail, kbench, bench rows vs legacy tail
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r3_b_tests.log 2>&1; rc=$?
tail -15 gpurun_out/r3_b_tests.log
[ $rc -eq 0 ] || exit $rc
python tools/kbench.py 2>&1 | tail -1 | tee gpurun_out/r3_b_kbench.log
python bench.py --steps 100 --no-cpu-baseline > gpurun_out/r3_b_bench_rows.json 2> gpurun_out/r3_b_bench_rows.err || exit 1
python bench.py --steps 100 --no-cpu-baseline --legacy-tail > gpurun_out/r3_b_bench_legacy.json 2> gpurun_out/r3_b_bench_legacy.err || exit 1
python bench.py --steps 100 --no-cpu-baseline --scan-cus 224 > gpurun_out/r3_b_bench_rows224.json 2> gpurun_out/r3_b_bench_rows224.err || exit 1
python bench.py --steps 100 --no-cpu-baseline --in-flight 2 --scan-cus 224 > gpurun_out/r3_b_bench_rows224_if2.json 2> /dev/null || exit 1
python - <<'PY'
import json
for f in ("rows", "legacy", "rows224", "rows224_if2"):
    d = json.load(open(f"gpurun_out/r3_b_bench_{f}.json")); r = d["roofline"]
    print(f, d["value"], d["ms_per_step"], r["avg_kernel_ms"], r["frac"], r.get("frac_whole_step"), r.get("isolated_avg_kernel_ms"), d["work"], d["phases_ms_per_step"])
PY
