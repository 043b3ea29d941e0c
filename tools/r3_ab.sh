# A/B on one box: a library built from another commit (tools/bin/libfocr_hip_base.so) against this tree's, alternating —
# tools/kbench.py (every kernel alone on the chip, exact sizes) and the default bench line
set -o pipefail
mkdir -p gpurun_out
for rep in 1 2 3; do
  for which in base new; do
    if [ $which = base ]; then export FOCR_HIP_LIB=$PWD/tools/bin/libfocr_hip_base.so; else unset FOCR_HIP_LIB; fi
    echo "$which: $(python tools/kbench.py 2>&1 | tail -1 | cut -c1-60)"
    timeout -k 10 120 python bench.py --no-cpu-baseline --no-e2e --steps 60 "$@" > gpurun_out/ab.json 2>/dev/null
    python -c "import json;d=json.load(open('gpurun_out/ab.json'));print('$which bench:', d['value'], d['ms_per_step'], d['roofline']['avg_kernel_ms'])"
  done
done
