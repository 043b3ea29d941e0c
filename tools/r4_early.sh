#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r04
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "pipeline or prefetch or fleet or three_contexts or pinned" 2>&1 | tail -5 || exit 1
FOCR_FUZZ_SECONDS=40 timeout -k 10 200 python -m pytest tests/test_gpu_fuzz_long.py -m gpu -x -q -s 2>&1 | tail -3 || exit 1
out=gpurun_out/r04/e2e_early.jsonl
rm -f $out
for rep in 1 2; do
  for n in 3 4; do
    timeout -k 10 120 python tools/e2e_leg.py --in-flight $n --steps 240 >> $out 2>/dev/null || exit 1
    timeout -k 10 120 python tools/e2e_leg.py --in-flight $n --steps 240 --resident >> $out 2>/dev/null || exit 1
  done
done
cat $out
