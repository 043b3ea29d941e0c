# 20 steps with and without the gather path (one rank, --force-gather), alternating on one box
python3 bench.py --no-cpu-baseline --no-e2e --steps 100 > /dev/null 2>&1
for rep in 1 2 3; do for g in "" "--force-gather"; do
python3 bench.py --no-cpu-baseline --no-e2e --steps 20 --warmup 5 $g 2> gpurun_out/k20g.err | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('20 steps [$g]:', d['value'], d['ms_per_step'], 'scan', d['roofline']['avg_kernel_ms'])"
grep "gather thread per gather" gpurun_out/k20g.err | cut -c1-330
done; done
