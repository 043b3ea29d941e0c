#!/bin/bash
# Rehearsal of bench.py's N-rank path on a one-GPU box: 2 ranks, both on device 0 (FOCR_BENCH_SHARE_GPU).
set -o pipefail
mkdir -p gpurun_out/r04
export FOCR_BENCH_SHARE_GPU=1 FOCR_BENCH_INIT_TIMEOUT=60
timeout -k 10 240 python bench.py --gpus 2 --steps 30 --warmup 6 --no-cpu-baseline > gpurun_out/r04/share2_c2.json 2> gpurun_out/r04/share2_c2.err
echo "rc=$?"
tail -5 gpurun_out/r04/share2_c2.err
cat gpurun_out/r04/share2_c2.json | cut -c1-600
