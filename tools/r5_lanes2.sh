#!/bin/bash
# round 5: lanes x depth again, with the lighter statistics (the lanes have slack now): 3x2 (product), 4x2, 4x1, 3x3
mkdir -p gpurun_out/r05; o=gpurun_out/r05
for i in 1 2 3 4; do
  for cfg in "3 2" "4 2" "4 1" "5 1"; do
    set -- $cfg
    python bench.py --steps 300 --no-cpu-baseline --no-e2e --no-extra-legs --in-flight $1 --depth $2 > $o/lanes_$1x$2.json 2>/dev/null
    python -c "
import json; d=json.load(open('$o/lanes_$1x$2.json')); st=d['step_stats']; print('lanes $1 depth $2: value', d['value'], 'ms', d['ms_per_step'], 'scan', d['roofline']['avg_kernel_ms'], 'dev p50/max', st['device_interval_ms_p50'], st['device_interval_ms_max'])"
  done
done
