# round 3, GPU call A: parity suite, then kernel-level and bench-level numbers of the column-drop build
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r3_a_tests.log 2>&1; rc=$?
tail -15 gpurun_out/r3_a_tests.log
[ $rc -eq 0 ] || exit $rc
python tools/kbench.py 2>&1 | tail -1 | tee gpurun_out/r3_a_kbench.log
python bench.py --steps 100 --no-cpu-baseline > gpurun_out/r3_a_bench_drop.json 2> gpurun_out/r3_a_bench_drop.err || exit 1
python bench.py --steps 100 --no-cpu-baseline --no-column-drop > gpurun_out/r3_a_bench_full.json 2> gpurun_out/r3_a_bench_full.err || exit 1
python - <<'PY'
import json
for f in ("drop", "full"):
    d = json.load(open(f"gpurun_out/r3_a_bench_{f}.json")); r = d["roofline"]
    print(f, d["value"], d["ms_per_step"], r["avg_kernel_ms"], r["frac"], r.get("frac_whole_step"), r.get("isolated_avg_kernel_ms"), d["work"], d["phases_ms_per_step"])
PY
