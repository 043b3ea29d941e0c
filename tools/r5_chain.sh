#!/bin/bash
# round 5: the statistics turn chain on / off, alternating on one box (plain 300-step runs: the protocol in which the slow rhythm shows up)
o=gpurun_out/r05; mkdir -p $o
for i in 1 2 3 4 5 6 7 8; do
  for m in chain nochain; do
    if [ $m = nochain ]; then export FOCR_NO_STATS_CHAIN=1; else unset FOCR_NO_STATS_CHAIN; fi
    python bench.py --steps 300 --no-cpu-baseline --no-e2e --no-extra-legs > $o/ch.json 2>/dev/null
    python -c "
import json; d=json.load(open('$o/ch.json')); st=d['step_stats']; print('$m', d['value'], d['ms_per_step'], d['roofline']['avg_kernel_ms'], 'dev p50', st['device_interval_ms_p50'], d['phases_ms_per_step'])"
  done
done
