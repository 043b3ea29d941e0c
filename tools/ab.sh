# A/B of experiment builds of the HIP library (font_ocr_amd/lib/exp/libfocr_hip_<X>.so) with the bench's default run
for v in ${AB_LIBS:-A B A B}; do
for q in ${AB_QUEUES:-8}; do
GPU_MAX_HW_QUEUES=$q FOCR_HIP_LIB=$PWD/font_ocr_amd/lib/exp/libfocr_hip_$v.so python bench.py --no-cpu-baseline --steps 100 ${AB_ARGS} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$v', 'queues $q', d['value'], d['ms_per_step'], r['avg_kernel_ms'], r['frac'], r.get('frac_whole_step'))"
done; done
