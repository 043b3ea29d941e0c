# configs[2] over the scan kernel's CU count and the batches in flight (one box, two rounds)
for rep in 1 2; do
for cfg in "224 3" "240 3" "256 3" "208 3" "224 4" "240 4" "224 2"; do
set -- $cfg
python bench.py --config c3 --pages-per-gpu 64 --steps 12 --warmup 2 --no-cpu-baseline --no-e2e --scan-cus $1 --in-flight $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('c3 cus', $1, 'in flight', $2, 'value', d['value'], 'ms', d['ms_per_step'], 'scan launch ms', r['avg_kernel_ms'], 'frac', r['frac'], 'whole step', r.get('frac_whole_step'))"
done
done
