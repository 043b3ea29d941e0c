#!/bin/bash
# round 5: the stall-tolerance pair of tests/test_gpu_executor_stall.py, several times, with the whole step_stats
o=gpurun_out/r05; mkdir -p $o
for i in 1 2 3 4; do for s in 0 3; do
  python bench.py --steps 150 --warmup 6 --no-cpu-baseline --no-e2e --no-extra-legs --settle-s 0.4 --inject-stall-ms $s > $o/st_${i}_$s.json 2> $o/st_${i}_$s.err
  python - $o/st_${i}_$s.json $s <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); st=d["step_stats"]; st.pop("note")
print("stall",sys.argv[2],d["value"],st)
PY
done; done
