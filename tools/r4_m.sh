set -o pipefail
repo=$PWD; out=$PWD/gpurun_out/r04; mkdir -p $out
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "estimates or full_size or fuzz or low_threshold or c3_geometry" > $out/m_tests.log 2>&1; rc=$?; tail -4 $out/m_tests.log; [ $rc -eq 0 ] || exit $rc
show() { python3 -c "import json,sys; d=json.load(open(sys.argv[1])); r=d['roofline']; print(sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'], r['avg_kernel_ms'], r.get('isolated_avg_kernel_ms'), r['frac'], d['size_estimates']['batches_redone_exact'])" $1; }
for i in 1 2 3; do python3 bench.py --no-cpu-baseline --no-e2e --steps 300 > $out/m_$i.json 2> $out/m_$i.err || tail -3 $out/m_$i.err; show $out/m_$i.json; done
python3 bench.py --no-cpu-baseline --no-e2e --config c3 --pages-per-gpu 64 --steps 12 --warmup 2 > $out/m_c3.json 2> /dev/null; show $out/m_c3.json
bash tools/timeline.sh > $out/m_timeline.log 2>&1; head -30 $out/m_timeline.log
