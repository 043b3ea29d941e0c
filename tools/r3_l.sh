set -o pipefail
mkdir -p gpurun_out
S=$PWD/font_ocr_amd/lib/exp/libfocr_hip_small.so
FOCR_HIP_LIB=$S python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "full_size_c2 or c2_page0 or fuzz" > gpurun_out/r3_l_tests.log 2>&1 || { tail -5 gpurun_out/r3_l_tests.log; exit 1; }
tail -2 gpurun_out/r3_l_tests.log
for cus in 224 240 248 256; do
  FOCR_HIP_LIB=$S python bench.py --steps 100 --no-cpu-baseline --no-e2e --scan-cus $cus > gpurun_out/r3_l_small_cu$cus.json 2> /dev/null || exit 1
done
FOCR_HIP_LIB=$S python bench.py --steps 100 --no-cpu-baseline --no-e2e --scan-cus 256 --in-flight 2 > gpurun_out/r3_l_small_cu256_if2.json 2> /dev/null || exit 1
FOCR_HIP_LIB=$S python bench.py --steps 100 --no-cpu-baseline --no-e2e --scan-cus 256 --in-flight 4 > gpurun_out/r3_l_small_cu256_if4.json 2> /dev/null || exit 1
python bench.py --steps 100 --no-cpu-baseline --no-e2e > gpurun_out/r3_l_base_cu224.json 2> /dev/null || exit 1
python bench.py --steps 100 --no-cpu-baseline --no-e2e --scan-cus 256 > gpurun_out/r3_l_base_cu256.json 2> /dev/null || exit 1
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r3_l_*.json")):
    d = json.load(open(f)); r = d["roofline"]
    print(f.split("r3_l_")[1], d["value"], d["ms_per_step"], r["avg_kernel_ms"], r["frac"], r.get("frac_whole_step"), d["phases_ms_per_step"])
PY
