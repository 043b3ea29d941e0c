#!/bin/bash
# round 5: cheap knobs under the new executor: tail grids, depth 3
o=gpurun_out/r05; mkdir -p $o
run() { tag=$1; shift; "$@" > $o/k_$tag.json 2> $o/k_$tag.err; python -c "
import json; d=json.load(open('$o/k_$tag.json')); print('$tag', d['value'], d['ms_per_step'], d['roofline']['avg_kernel_ms'], d['phases_ms_per_step'])"; }
B="python bench.py --steps 300 --no-cpu-baseline --no-e2e --no-extra-legs"
run base $B
FOCR_BENCH_TAIL_GRID=1/2 run grid_half $B
FOCR_BENCH_TAIL_GRID=3/2 run grid_1p5 $B
FOCR_BENCH_TAIL_GRID=2/1 run grid_2 $B
run depth3 $B --depth 3
run base2 $B
