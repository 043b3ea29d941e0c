set -o pipefail
repo=$PWD; out=$PWD/gpurun_out/r04; mkdir -p $out
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "pipeline or contexts or c2_page0 or small_batch or soak" > $out/l_tests.log 2>&1; rc=$?; tail -4 $out/l_tests.log; [ $rc -eq 0 ] || exit $rc
show() { python3 -c "import json,sys; d=json.load(open(sys.argv[1])); r=d['roofline']; print(sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'], r['avg_kernel_ms'], r.get('isolated_avg_kernel_ms'), r['frac'])" $1; }
for i in 1 2 3; do
python3 bench.py --no-cpu-baseline --no-e2e --steps 300 > $out/l_sig_$i.json 2> $out/l_sig_$i.err || tail -3 $out/l_sig_$i.err; show $out/l_sig_$i.json
FOCR_SCAN_HANDOVER=event python3 bench.py --no-cpu-baseline --no-e2e --steps 300 > $out/l_ev_$i.json 2> $out/l_ev_$i.err; show $out/l_ev_$i.json
done
python3 bench.py --no-cpu-baseline --no-e2e --steps 300 --in-flight 4 > $out/l_sig_4.json 2> /dev/null; show $out/l_sig_4.json
python3 bench.py --no-cpu-baseline --no-e2e --steps 300 --scan-cus 232 > $out/l_sig_232.json 2> /dev/null; show $out/l_sig_232.json
python3 bench.py --no-cpu-baseline --no-e2e --steps 300 --scan-cus 240 > $out/l_sig_240.json 2> /dev/null; show $out/l_sig_240.json
python3 bench.py --no-cpu-baseline --no-e2e --config c3 --pages-per-gpu 64 --steps 12 --warmup 2 > $out/l_c3_sig.json 2> /dev/null; show $out/l_c3_sig.json
FOCR_SCAN_HANDOVER=event python3 bench.py --no-cpu-baseline --no-e2e --config c3 --pages-per-gpu 64 --steps 12 --warmup 2 > $out/l_c3_ev.json 2> /dev/null; show $out/l_c3_ev.json
bash tools/timeline.sh > $out/l_timeline.log 2>&1; head -12 $out/l_timeline.log
