#!/bin/bash
# round 5: kernel timelines of several identical runs — the pipeline settles into one of two steady states (33 / 31.5 Gpx/s)
o=gpurun_out/r05; mkdir -p $o
for i in 1 2 3 4 5; do
  bash tools/timeline.sh > $o/mode_$i.log 2>&1
  python -c "
import json; d=json.load(open('gpurun_out/timeline_run.json')); print('run $i value', d['value'], d['ms_per_step'], d['roofline']['avg_kernel_ms'], d['phases_ms_per_step'])"
  head -16 $o/mode_$i.log | cut -c1-110
done
