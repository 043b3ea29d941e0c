#!/bin/bash
# the PCIe-inclusive leg at 3 and 4 batches in flight, alternating with the resident leg of the same loop (one box)
set -o pipefail
mkdir -p gpurun_out/r04
out=gpurun_out/r04/e2e_inflight.jsonl
rm -f $out
for rep in 1 2; do
  for n in 3 4; do
    timeout -k 10 120 python tools/e2e_leg.py --in-flight $n --steps 240 >> $out 2>/dev/null || exit 1
    timeout -k 10 120 python tools/e2e_leg.py --in-flight $n --steps 240 --resident >> $out 2>/dev/null || exit 1
  done
done
cat $out
