for thr in 0.8 0.9 0.97 1.5; do
python bench.py --no-cpu-baseline --steps 60 --threshold $thr 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('thr', $thr, d['value'], d['ms_per_step'], r['avg_kernel_ms'], 'gap', round(d['ms_per_step']-r['avg_kernel_ms'],3), r.get('isolated_avg_kernel_ms'), d['work'])"
done
