# final sanity of bench.py's other configurations after the late bench.py changes
mkdir -p gpurun_out/r04
run() { tag=$1; shift; python3 bench.py --no-cpu-baseline --no-e2e "$@" > gpurun_out/r04/final_$tag.json 2> gpurun_out/r04/final_$tag.err || { echo "$tag FAILED"; tail -3 gpurun_out/r04/final_$tag.err; return; }
python3 -c "
import json; d=json.load(open('gpurun_out/r04/final_$tag.json')); print('$tag', d['value'], d['ms_per_step'], d['roofline']['frac'], d['size_estimates']['batches_redone_exact'], d['host_gc']['longest_pause_ms'], d.get('optional_leg_errors'))"; }
run c3 --config c3 --pages-per-gpu 64 --steps 12 --warmup 2
run c3_stream --config c3 --pages-per-gpu 64 --c3-pages 1024 --steps 2 --warmup 1
run c4 --config c4 --c4-pages 2048 --steps 3 --warmup 1
run gather --force-gather --steps 300
run gather20 --force-gather --steps 20 --warmup 5
run noise --noise --steps 100
run serial --in-flight 1 --steps 40
