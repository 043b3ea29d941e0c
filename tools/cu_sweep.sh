set -e
for inf in 3 4; do for cus in 176 192 208 224 240 256; do
python bench.py --steps 60 --warmup 6 --no-cpu-baseline --scan-cus $cus --in-flight $inf > gpurun_out/sw_${inf}_${cus}.json 2>/dev/null
done; done
python - <<PY
import json
for inf in (3,4):
    for cus in (176,192,208,224,240,256):
        d=json.loads(open("gpurun_out/sw_%d_%d.json"%(inf,cus)).read().strip().splitlines()[-1])
        r=d["roofline"]
        print(inf,cus,d["value"],d["ms_per_step"],r["avg_kernel_ms"],r["frac"],r.get("frac_whole_step"))
PY
