# how often does a 20-step timed region lose milliseconds to a hiccup?  N runs each at three and four batches in flight, alternating
python3 bench.py --no-cpu-baseline --no-e2e --steps 100 > /dev/null 2>&1
for rep in $(seq 1 ${1:-12}); do for n in 3 4; do
python3 bench.py --no-cpu-baseline --no-e2e --steps 20 --warmup 5 --in-flight $n 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['phases_ms_per_step']; print('in flight $n:', d['value'], d['ms_per_step'], 'scan', d['roofline']['avg_kernel_ms'], 'phases', p['stats'], p['scan'], p['verify'], p['order'], p['process_hits'])"
done; done
