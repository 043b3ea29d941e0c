repo=$PWD; out=$PWD/gpurun_out/r04; mkdir -p $out
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "pipeline or contexts" > $out/g_tests.log 2>&1; echo "[r4_g] tests rc=$?"; tail -5 $out/g_tests.log
show() { python3 -c "import json,sys; d=json.load(open(sys.argv[1])); r=d['roofline']; print(sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'], r['avg_kernel_ms'], r.get('isolated_avg_kernel_ms'), d['config'].get('cu_partition'), d.get('e2e_value_incl_h2d_pipelined'))" $1; }
for sp in 0 4 5 6 0 4 5; do python3 bench.py --no-cpu-baseline --steps 200 --cu-split $sp > $out/g_c2_split$sp.json 2> $out/g_c2_split$sp.err; show $out/g_c2_split$sp.json; done
for sp in 4 5; do python3 bench.py --no-cpu-baseline --no-e2e --steps 200 --cu-split $sp --in-flight 4 > $out/g_c2_split${sp}_4.json 2> $out/g_c2_split${sp}_4.err; show $out/g_c2_split${sp}_4.json; done
for sp in 0 4 6; do python3 bench.py --no-cpu-baseline --no-e2e --config c3 --pages-per-gpu 64 --steps 12 --warmup 2 --cu-split $sp > $out/g_c3_split$sp.json 2> $out/g_c3_split$sp.err; show $out/g_c3_split$sp.json; done
