# 20-step (the driver's K) and 300-step values: base library (tools/bin/libfocr_hip_base.so) against this tree, alternating
set -o pipefail
for rep in 1 2 3 4 5; do
  for which in base new; do
    for k in 20 300; do
      if [ $which = base ]; then export FOCR_HIP_LIB=$PWD/tools/bin/libfocr_hip_base.so; else unset FOCR_HIP_LIB; fi
      timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-e2e --steps $k --warmup 5 "$@" > gpurun_out/ab.json 2>/dev/null || { echo "$which $k failed"; continue; }
      python3 -c "import json;d=json.load(open('gpurun_out/ab.json'));print('$which steps $k:', d['value'], d['ms_per_step'], d['roofline']['avg_kernel_ms'])"
    done
  done
done
