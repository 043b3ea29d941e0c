# 20 and 300 steps at three to six batches in flight, alternating on one box (first pass warms the box and is not printed)
python3 bench.py --no-cpu-baseline --no-e2e --steps 100 > /dev/null 2>&1
for rep in 1 2; do for n in 3 4 5 6; do for k in 20 300; do
python3 bench.py --no-cpu-baseline --no-e2e --steps $k --warmup 5 --in-flight $n 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('in flight $n steps $k:', d['value'], d['ms_per_step'], 'scan', d['roofline']['avg_kernel_ms'], 'frac', d['roofline']['frac'], 'whole', d['roofline']['frac_whole_step'])"
done; done; done
