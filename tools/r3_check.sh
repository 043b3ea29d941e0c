# round 3: the validation set run before a commit that touches kernels (GPU box, repo root): parity suite, smoke, per-kernel table,
# default bench line, the RCCL gather path with one rank, the refusal of --gpus 2 on a one-GPU box
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r3_check_tests.log 2>&1; rc=$?
tail -3 gpurun_out/r3_check_tests.log
[ $rc -eq 0 ] || exit $rc
python __graft_entry__.py smoke 2>&1 | tail -1
bash tools/kprof.sh r3_check | head -22
python bench.py > gpurun_out/r3_check_bench_default.json 2> gpurun_out/r3_check_bench_default.err || { tail -5 gpurun_out/r3_check_bench_default.err; exit 1; }
python bench.py --force-gather --steps 40 --no-cpu-baseline > gpurun_out/r3_check_bench_force_gather.json 2> gpurun_out/r3_check_fg.err || { tail -5 gpurun_out/r3_check_fg.err; exit 1; }
python bench.py --gpus 2 --steps 5 > /dev/null 2> gpurun_out/r3_check_gpus2.err; echo "bench --gpus 2 on this box: rc=$? ($(tail -1 gpurun_out/r3_check_gpus2.err))"
python - <<'PY'
import json
for f in ("default", "force_gather"):
    d = json.load(open(f"gpurun_out/r3_check_bench_{f}.json")); r = d["roofline"]
    print(f, d["value"], d["ms_per_step"], d["n_gpus"], d["ranks_seen"], d.get("per_rank_value"), r["avg_kernel_ms"], r["frac"], r.get("frac_whole_step"), r.get("isolated_avg_kernel_ms"), r.get("traffic"),
          d.get("e2e_value_incl_h2d_pipelined"), d.get("cpu_baseline", {}).get("value"), d.get("parity"), d.get("size_estimates", {}).get("batches_redone_exact"))
PY
