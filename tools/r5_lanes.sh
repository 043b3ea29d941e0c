#!/bin/bash
# round 5: independent streams — lanes x depth 1 against 3 x 2
o=gpurun_out/r05; mkdir -p $o
for rep in 1 2; do for cfg in "3 2" "4 1" "5 1" "6 1" "3 1"; do set -- $cfg
  python bench.py --steps 300 --no-cpu-baseline --no-e2e --no-extra-legs --in-flight $1 --depth $2 > $o/ln_$1_$2.json 2>/dev/null
  python -c "
import json; d=json.load(open('$o/ln_$1_$2.json')); st=d['step_stats']; print('lanes $1 depth $2', d['value'], d['ms_per_step'], d['roofline']['avg_kernel_ms'], 'dev p50/max', st['device_interval_ms_p50'], st['device_interval_ms_max'], d['phases_ms_per_step']['total'])"
done; done
