#!/bin/bash
# round 5: the gather path against the number of hardware queues the runtime may use (an executor has 3 lane + copy + side streams)
o=gpurun_out/r05; mkdir -p $o
for q in 8 12 16 24; do
  GPU_MAX_HW_QUEUES=$q python bench.py --steps 100 --no-cpu-baseline --no-e2e --no-extra-legs --force-gather > $o/fg_$q.json 2> $o/fg_$q.err
  python -c "
import json; d=json.load(open('$o/fg_$q.json')); print('queues $q force-gather', d['value'])"
done
GPU_MAX_HW_QUEUES=16 python bench.py --steps 100 --no-cpu-baseline --no-extra-legs > $o/q16.json 2>/dev/null
python -c "
import json; d=json.load(open('$o/q16.json')); print('queues 16 plain', d['value'], d['e2e_value_incl_h2d_pipelined'])"
