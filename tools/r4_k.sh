set -o pipefail
repo=$PWD; out=$PWD/gpurun_out/r04; mkdir -p $out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/k_tests.log 2>&1; rc=$?; tail -4 $out/k_tests.log; [ $rc -eq 0 ] || exit $rc
python __graft_entry__.py smoke 2>&1 | tail -1
show() { python3 -c "import json,sys; d=json.load(open(sys.argv[1])); r=d['roofline']; print(sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'], r['avg_kernel_ms'], r.get('frac'), r.get('frac_issued'), d.get('e2e_value_incl_h2d_pipelined'), d['size_estimates']['batches_redone_exact'], d.get('host_gc'))" $1; }
python3 bench.py --no-cpu-baseline --steps 300 > $out/k_c2.json 2> $out/k_c2.err; show $out/k_c2.json
for cu in 216 224 232 240; do python3 bench.py --no-cpu-baseline --no-e2e --config c3 --pages-per-gpu 64 --steps 12 --warmup 2 --scan-cus $cu > $out/k_c3_cus$cu.json 2> $out/k_c3_cus$cu.err; echo "c3 scan cus $cu"; show $out/k_c3_cus$cu.json; done
python3 bench.py --no-cpu-baseline --no-e2e --config c3 --pages-per-gpu 64 --c3-pages 1024 --steps 2 --warmup 1 > $out/k_c3_stream.json 2> $out/k_c3_stream.err; show $out/k_c3_stream.json
python3 bench.py --no-cpu-baseline --no-e2e --config c3 --pages-per-gpu 64 --steps 12 --warmup 2 --in-flight 4 > $out/k_c3_4.json 2> $out/k_c3_4.err; echo "c3 4 in flight"; show $out/k_c3_4.json
