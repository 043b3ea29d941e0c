#!/bin/bash
# round 5: the executor (lanes x depth, one enqueue thread) under the driver's command, at 300 steps, with injected host stalls, and at depth 1
set -e
o=gpurun_out/r05
mkdir -p $o
python bench.py --gpus 1 --steps 20 --warmup 5 > $o/drv1.json 2> $o/drv1.err
for i in 2 3 4; do python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-e2e --no-extra-legs > $o/drv$i.json 2> $o/drv$i.err; done
python bench.py --steps 300 --no-cpu-baseline --no-e2e --no-extra-legs > $o/s300.json 2> $o/s300.err
python bench.py --steps 300 --no-cpu-baseline --no-e2e --no-extra-legs --depth 1 > $o/s300_d1.json 2> $o/s300_d1.err
python bench.py --steps 200 --no-cpu-baseline --no-e2e --no-extra-legs --inject-stall-ms 3 > $o/stall3.json 2> $o/stall3.err
python bench.py --steps 200 --no-cpu-baseline --no-e2e --no-extra-legs --inject-stall-ms 3 --depth 1 > $o/stall3_d1.json 2> $o/stall3_d1.err
python bench.py --steps 200 --no-cpu-baseline --no-e2e --no-extra-legs --inject-stall-ms 6 > $o/stall6.json 2> $o/stall6.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-e2e --no-extra-legs --depth 1 > $o/drv_d1.json 2> $o/drv_d1.err
for f in drv1 drv2 drv3 drv4 s300 s300_d1 stall3 stall3_d1 stall6 drv_d1; do python - $o/$f.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(sys.argv[1].split('/')[-1], d["value"], d["ms_per_step"], d["roofline"]["avg_kernel_ms"], d.get("step_stats"))
PY
done
