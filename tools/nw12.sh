for v in A B; do for cus in 192 224 256; do
GPU_MAX_HW_QUEUES=8 FOCR_HIP_LIB=$PWD/font_ocr_amd/lib/exp/libfocr_hip_$v.so python bench.py --no-cpu-baseline --steps 60 --scan-cus $cus 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$v', $cus, d['value'], d['ms_per_step'], r['kernel'], r['avg_kernel_ms'], r['frac'], r.get('frac_whole_step'), r.get('isolated_avg_kernel_ms'))"
done; done
