#!/bin/bash
# round 5: FETCH_SIZE of the scan kernel (one --pmc pass over tools/kbench.py) and the A/B against tools/bin/libfocr_hip_base.so
mkdir -p gpurun_out/r05; out=$PWD/gpurun_out/r05; repo=$PWD
for which in base new; do
  if [ $which = base ]; then export FOCR_HIP_LIB=$repo/tools/bin/libfocr_hip_base.so; else unset FOCR_HIP_LIB; fi
  (cd /tmp && TMPDIR=/tmp rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/tmp_f_$which -o p -- python3 $repo/tools/kbench.py > $out/fetch_$which.log 2>&1)
  f=$(find $out/tmp_f_$which -name "*counter_collection.csv" | head -1)
  python3 - "$f" $which <<'PY'
import csv, sys
v = [float(x["Counter_Value"]) for x in csv.DictReader(open(sys.argv[1])) if "scan_mfma2s" in x["Kernel_Name"] and x["Counter_Name"] == "FETCH_SIZE"]
print(sys.argv[2], "scan kernel FETCH_SIZE KB per launch", round(sum(v) / len(v), 1), "launches", len(v))
PY
  rm -rf $out/tmp_f_$which
done
unset FOCR_HIP_LIB
bash tools/r4_ab.sh
