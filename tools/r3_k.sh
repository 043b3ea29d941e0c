set -o pipefail
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r3_k_tests.log 2>&1; rc=$?
tail -5 gpurun_out/r3_k_tests.log
[ $rc -eq 0 ] || exit $rc
python __graft_entry__.py smoke 2>&1 | tail -2
bash tools/kprof.sh r3_k_rows
python bench.py > gpurun_out/r3_k_bench_default.json 2> gpurun_out/r3_k_bench_default.err || { tail -5 gpurun_out/r3_k_bench_default.err; exit 1; }
python bench.py --steps 100 --no-cpu-baseline --no-e2e --scan-cus 208 > gpurun_out/r3_k_bench_cu208.json 2> /dev/null || exit 1
python bench.py --steps 100 --no-cpu-baseline --no-e2e --scan-cus 224 > gpurun_out/r3_k_bench_cu224.json 2> /dev/null || exit 1
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r3_k_bench*.json")):
    d = json.load(open(f)); r = d["roofline"]
    print(f.split("r3_k_")[1], d["value"], d["ms_per_step"], r["avg_kernel_ms"], r["frac"], r.get("frac_whole_step"), r.get("isolated_avg_kernel_ms"), r.get("traffic"), d.get("e2e_value_incl_h2d_pipelined"), d.get("parity"), d["phases_ms_per_step"])
PY
