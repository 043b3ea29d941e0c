#!/bin/bash
# round 5: identical 300-step runs until one falls into the slow rhythm (value below 0.97 of the best so far); its schedule is printed
mkdir -p gpurun_out/r05; o=gpurun_out/r05; best=0
for i in $(seq 1 14); do
  FOCR_BENCH_DUMP_TICKETS=$o/hunt_$i.json python bench.py --steps 300 --no-cpu-baseline --no-e2e --no-extra-legs > $o/hunt_run_$i.json 2>/dev/null
  v=$(python -c "import json; print(json.load(open('$o/hunt_run_$i.json'))['value'])")
  echo "run $i value $v"
  best=$(python -c "print(max($best, $v))")
  if python -c "import sys; sys.exit(0 if $v < 0.97 * $best else 1)"; then
    echo "== slow run $i"; python tools/r5_phase_view.py $o/hunt_$i.json 150 24; cp $o/hunt_$i.json $o/slow_tickets.json; break
  fi
done
