#!/bin/bash
# round 5: the PCIe-inclusive leg alone (announced / not announced / resident), then traced with its memory copies
o=gpurun_out/r05; mkdir -p $o; repo=$PWD
for v in "" "--no-prefetch" "--resident" "" "--resident"; do python3 tools/e2e_leg.py --steps 240 $v; done
(cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $repo/$o/tmp_e2e -o t -- python3 $repo/tools/e2e_leg.py --steps 120 > $repo/$o/e2e_traced_run.json 2> $repo/$o/e2e_traced.err)
for k in kernel_trace memory_copy_trace; do f=$(find $o/tmp_e2e -name "*${k}.csv" | head -1); [ -n "$f" ] && cp $f $o/e2e_${k}.csv; done; rm -rf $o/tmp_e2e
python3 tools/e2e_timeline.py $o/e2e_kernel_trace.csv $o/e2e_memory_copy_trace.csv | head -40
rm -f $o/e2e_kernel_trace.csv $o/e2e_memory_copy_trace.csv
