mkdir -p gpurun_out
for rep in 1 2 3 4 5 6; do
  for which in base new; do
    if [ $which = base ]; then export FOCR_HIP_LIB=$PWD/tools/bin/libfocr_hip_base.so; else unset FOCR_HIP_LIB; fi
    python3 bench.py --no-cpu-baseline --no-e2e --no-extra-legs --steps 300 > gpurun_out/ab.json 2>/dev/null
    python3 bench.py --no-cpu-baseline --no-e2e --no-extra-legs --gpus 1 --steps 20 --warmup 5 > gpurun_out/ab20.json 2>/dev/null
    python3 -c "import json;d=json.load(open('gpurun_out/ab.json'));e=json.load(open('gpurun_out/ab20.json'));print('$which: 300 steps', d['value'], d['ms_per_step'], d['roofline']['avg_kernel_ms'], '| 20 steps', e['value'], e['ms_per_step'])"
  done
done
