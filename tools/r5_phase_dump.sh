#!/bin/bash
# round 5: eight identical runs with every batch's phases on the device's clock (bench.py FOCR_BENCH_DUMP_TICKETS + focr_debug_phase_stamps), and
# a window of each run's schedule (tools/r5_phase_view.py): the two rhythms of DESIGN.md section 5 without a profiler in the process
mkdir -p gpurun_out/r05
bash tools/r5_modes2.sh > gpurun_out/r05/modes2b.log 2>&1
grep "^run" gpurun_out/r05/modes2b.log | cut -c1-60
for i in 1 2 3 4 5 6 7 8; do echo "== run $i"; python tools/r5_phase_view.py gpurun_out/r05/tk_$i.json 150 14; done
