#!/bin/bash
# round 5: the executor's front stream (statistics ahead of the lanes, paced by the scan turns) against the lanes doing their own (FOCR_PIPE_NO_FRONT=1), same lease
mkdir -p gpurun_out/r05; o=gpurun_out/r05
run() {  # name, env...
  local name=$1; shift
  env "$@" FOCR_BENCH_DUMP_TICKETS=$o/front_${name}.tickets.json python bench.py --steps 300 --no-cpu-baseline --no-e2e --no-extra-legs > $o/front_${name}.json 2>/dev/null
  env "$@" python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-e2e --no-extra-legs > $o/front_${name}_d.json 2>/dev/null
  python - <<PY
import json
a=json.load(open("$o/front_${name}.json")); b=json.load(open("$o/front_${name}_d.json"))
print("$name: 300 steps %.0f   20 steps %.0f   scan-alone frac %.3f" % (a["value"], b["value"], a["roofline"]["frac"]))
PY
}
for i in 1 2 3 4; do
  run lead2_$i FOCR_FRONT_LEAD=2
  run lead1_$i FOCR_FRONT_LEAD=1
  run lead3_$i FOCR_FRONT_LEAD=3
  run nofront_$i FOCR_PIPE_NO_FRONT=1
done
python tools/r5_phase_view.py $o/front_lead2_2.tickets.json 150 12
python tools/r5_phase_view.py $o/front_lead1_2.tickets.json 150 12
