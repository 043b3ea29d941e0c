#!/bin/bash
# round 5: lanes x scan CUs with the new executor (300 steps each, no optional legs)
o=gpurun_out/r05; mkdir -p $o
for lanes in 3 4; do for cus in 216 224 232 240 256; do
  python bench.py --steps 300 --no-cpu-baseline --no-e2e --no-extra-legs --in-flight $lanes --scan-cus $cus > $o/sw_${lanes}_${cus}.json 2> $o/sw_${lanes}_${cus}.err
  python - $o/sw_${lanes}_${cus}.json $lanes $cus <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); st=d["step_stats"]
print("lanes",sys.argv[2],"scan_cus",sys.argv[3],"value",d["value"],"ms",d["ms_per_step"],"scan_ms",d["roofline"]["avg_kernel_ms"],"dev_p50",st["device_interval_ms_p50"])
PY
done; done
