#!/bin/bash
# round 5: the driver's command ten times in a row on one lease (first and last with every optional leg, the rest without)
o=gpurun_out/r05; mkdir -p $o
for i in 1 2 3 4 5 6 7 8 9 10; do
  extra="--no-cpu-baseline --no-e2e --no-extra-legs"; if [ $i = 1 ] || [ $i = 10 ]; then extra=""; fi
  python3 bench.py --gpus 1 --steps 20 --warmup 5 $extra > $o/d10_$i.json 2> $o/d10_$i.err
  python -c "
import json; d=json.load(open('$o/d10_$i.json')); st=d['step_stats']; print('driver run $i', d['value'], d['ms_per_step'], d['roofline']['avg_kernel_ms'], 'frac_whole', d['roofline']['frac_whole_step'], 'first', st['first_completion_ms'], 'dev p50/max', st['device_interval_ms_p50'], st['device_interval_ms_max'], 'host gap', st['longest_host_gap_ms'], d.get('c3_value'), d.get('c4_stream_value'), d.get('e2e_value_incl_h2d_pipelined'))"
done
