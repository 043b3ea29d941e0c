set -o pipefail
repo=$PWD; out=$PWD/gpurun_out/r04; mkdir -p $out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/e_tests.log 2>&1; rc=$?; tail -5 $out/e_tests.log; [ $rc -eq 0 ] || exit $rc
python __graft_entry__.py smoke 2>&1 | tail -1
bash tools/kprof.sh r04_e_c2_hits KB_TAIL=1 > $out/e_kprof_c2_hits.log 2>&1; cat $out/e_kprof_c2_hits.log | head -24
bash tools/kprof.sh r04_e_c2_rows3 KB_TAIL=2 > $out/e_kprof_c2_rows3.log 2>&1; cat $out/e_kprof_c2_rows3.log | head -24
bash tools/kprof.sh r04_e_c3_hits KB_TAIL=1 KB_CONFIG=c3 > $out/e_kprof_c3_hits.log 2>&1; cat $out/e_kprof_c3_hits.log | head -24
show() { python3 -c "import json,sys; d=json.load(open(sys.argv[1])); print(sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'], d['roofline']['avg_kernel_ms'], d['size_estimates'])" $1; }
for t in hits rows3 hits rows3; do python3 bench.py --no-cpu-baseline --no-e2e --steps 200 --tail $t > $out/e_c2_$t.json 2> $out/e_c2_$t.err; show $out/e_c2_$t.json; done
for t in hits rows3; do python3 bench.py --no-cpu-baseline --no-e2e --config c3 --pages-per-gpu 64 --steps 12 --warmup 2 --tail $t > $out/e_c3_$t.json 2> $out/e_c3_$t.err; show $out/e_c3_$t.json; done
