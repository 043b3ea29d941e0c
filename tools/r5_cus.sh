#!/bin/bash
# round 5: scan CUs with the int16-plane scan kernel (alternating, one box)
o=gpurun_out/r05; mkdir -p $o
for rep in 1 2 3; do for cus in 208 216 224; do
  python bench.py --steps 300 --no-cpu-baseline --no-e2e --no-extra-legs --scan-cus $cus > $o/cu_$cus.json 2>/dev/null
  python -c "
import json; d=json.load(open('$o/cu_$cus.json')); st=d['step_stats']; print('scan_cus $cus', d['value'], d['ms_per_step'], d['roofline']['avg_kernel_ms'], 'dev p50/max', st['device_interval_ms_p50'], st['device_interval_ms_max'])"
done; done
