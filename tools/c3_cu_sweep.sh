for cus in 160 176 192 208 224; do
python bench.py --config c3 --pages-per-gpu 64 --steps 12 --warmup 2 --no-cpu-baseline --scan-cus $cus 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('c3', $cus, d['value'], d['ms_per_step'], r['avg_kernel_ms'], r['frac'], r.get('frac_whole_step'), d['kernels_ms_per_step'], d['work'])"
done
