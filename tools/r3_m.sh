set -o pipefail
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r3_m_tests.log 2>&1; rc=$?
tail -3 gpurun_out/r3_m_tests.log
[ $rc -eq 0 ] || exit $rc
bash tools/kprof.sh r3_m_rows | head -8
python bench.py --steps 100 --no-cpu-baseline --no-e2e > gpurun_out/r3_m_bench.json 2> /dev/null || exit 1
python - <<'PY'
import json
d = json.load(open("gpurun_out/r3_m_bench.json")); r = d["roofline"]
print(d["value"], d["ms_per_step"], r["avg_kernel_ms"], r["frac"], r.get("frac_whole_step"), d["phases_ms_per_step"])
PY
