# bench.py over the scan kernel's CU count and the batches in flight (tools; not part of the product)
set -e
for inf in ${SW_INFLIGHT:-3 4}; do for cus in ${SW_CUS:-176 192 208 224 240 256}; do
python bench.py --steps 60 --warmup 6 --no-cpu-baseline --scan-cus $cus --in-flight $inf 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print($inf, $cus, d['value'], d['ms_per_step'], r['avg_kernel_ms'], r['frac'], r.get('frac_whole_step'))"
done; done
