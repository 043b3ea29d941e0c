# the end-of-stream hint (focr_pipe_end_of_stream at bench.py's fence) on and off, 20 and 300 steps, alternating on one box
python3 bench.py --no-cpu-baseline --no-e2e --steps 100 > /dev/null 2>&1
for rep in 1 2 3 4 5; do for eos in on off; do for k in 20 300; do
if [ $eos = off ]; then export FOCR_BENCH_NO_EOS=1; else unset FOCR_BENCH_NO_EOS; fi
python3 bench.py --no-cpu-baseline --no-e2e --steps $k --warmup 5 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('hint $eos steps $k:', d['value'], d['ms_per_step'], d['roofline']['avg_kernel_ms'])"
done; done; done
