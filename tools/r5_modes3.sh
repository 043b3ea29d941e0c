#!/bin/bash
# round 5: catch the slow steady state under the kernel trace (the bench's usual protocol: settling rounds in front of the timed steps)
o=gpurun_out/r05; mkdir -p $o; repo=$PWD
for i in 1 2 3 4 5 6 7 8; do
  (cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --output-format csv -d $repo/$o/tmp_tl -o t -- python3 $repo/bench.py --no-cpu-baseline --no-e2e --no-extra-legs --steps 200 --warmup 6 --settle-s 0.3 > $repo/$o/tl_run_$i.json 2> $repo/$o/tl_$i.err)
  f=$(find $o/tmp_tl -name "*kernel_trace.csv" | head -1); v=$(python -c "
import json; d=json.load(open('$o/tl_run_$i.json')); print(d['value'])")
  echo "run $i value $v"
  python3 tools/timeline.py $f | head -3
  if python -c "import sys; sys.exit(0 if float('$v') < 31800 else 1)"; then cp $f $o/bad_trace_$i.csv; echo "kept bad trace $i"; fi
  rm -rf $o/tmp_tl
done
