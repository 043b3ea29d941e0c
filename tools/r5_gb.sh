#!/bin/bash
# round 5: bands per workgroup of the statistics kernel (FOCR_S8_GB = 1 .. 5; a workgroup = the page's strips x GB bands)
mkdir -p gpurun_out/r05; o=gpurun_out/r05
for i in 1 2 3 4; do
  for gb in 1 2 3 4 5; do
    FOCR_S8_GB=$gb python bench.py --steps 300 --no-cpu-baseline --no-e2e --no-extra-legs > $o/gb_$gb.json 2>/dev/null
    python -c "import json;d=json.load(open('$o/gb_$gb.json'));p=d['phases_ms_per_step'];print('GB $gb: value', d['value'], 'ms', d['ms_per_step'], 'scan', d['roofline']['avg_kernel_ms'], 'stats phase', round(p['stats'],3), 'scan phase', round(p['scan'],3), 'dev p50', d['step_stats']['device_interval_ms_p50'])"
  done
done
