#!/bin/bash
# round 5: identical 300-step runs; the per-ticket stamps of each are kept (gpurun_out/r05/tk_<i>.json)
o=gpurun_out/r05; mkdir -p $o
for i in 1 2 3 4 5 6 7 8; do
  FOCR_BENCH_DUMP_TICKETS=$o/tk_$i.json python bench.py --steps 300 --no-cpu-baseline --no-e2e --no-extra-legs > $o/md_$i.json 2>/dev/null
  python -c "
import json; d=json.load(open('$o/md_$i.json')); st=d['step_stats']; st.pop('note'); print('run $i', d['value'], d['ms_per_step'], d['roofline']['avg_kernel_ms'], d['phases_ms_per_step'], st)"
done
