set -o pipefail
repo=$PWD; out=$PWD/gpurun_out/r04; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "c3 or fuzz or random_banks" > $out/i_tests.log 2>&1; rc=$?; tail -4 $out/i_tests.log; [ $rc -eq 0 ] || exit $rc
bash tools/kprof.sh r04_i_c3_chunks KB_CONFIG=c3 > $out/i_kprof_c3.log 2>&1; head -12 $out/i_kprof_c3.log
show() { python3 -c "import json,sys; d=json.load(open(sys.argv[1])); r=d['roofline']; print(sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'], r['avg_kernel_ms'], d['size_estimates']['largest_page_row'])" $1; }
for v in chunks global chunks global; do
  if [ $v = global ]; then export FOCR_VERIFY_GLOBAL=1; else unset FOCR_VERIFY_GLOBAL; fi
  python3 bench.py --no-cpu-baseline --no-e2e --config c3 --pages-per-gpu 64 --steps 12 --warmup 2 > $out/i_c3_$v.json 2> $out/i_c3_$v.err; show $out/i_c3_$v.json
done
unset FOCR_VERIFY_GLOBAL
for cu in 208 216 232 240; do python3 bench.py --no-cpu-baseline --no-e2e --config c3 --pages-per-gpu 64 --steps 12 --warmup 2 --scan-cus $cu > $out/i_c3_cus$cu.json 2> $out/i_c3_cus$cu.err; echo "scan cus $cu"; show $out/i_c3_cus$cu.json; done
