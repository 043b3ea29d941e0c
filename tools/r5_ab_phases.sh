mkdir -p gpurun_out
for rep in 1 2 3; do
  for which in base new; do
    if [ $which = base ]; then export FOCR_HIP_LIB=$PWD/tools/bin/libfocr_hip_base.so; else unset FOCR_HIP_LIB; fi
    timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-e2e --no-extra-legs --steps 300 > gpurun_out/ab.json 2>/dev/null
    python3 -c "import json;d=json.load(open('gpurun_out/ab.json'));p=d['phases_ms_per_step'];print('$which:', d['value'], d['ms_per_step'], d['roofline']['avg_kernel_ms'], 'phases', {k: round(v,3) for k,v in p.items()}, 'work', d['work'])"
  done
done
