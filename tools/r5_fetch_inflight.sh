#!/bin/bash
# round 5: the scan kernel's FETCH_SIZE per launch INSIDE the pipeline (bench.py under rocprofv3 --pmc), the statistics kernel's own
# work list (default) against mark bytes + compact_live_tiles (FOCR_NO_STATS_APPEND=1)
mkdir -p gpurun_out/r05; out=$PWD/gpurun_out/r05; repo=$PWD
for which in append compact; do
  if [ $which = compact ]; then export FOCR_NO_STATS_APPEND=1; else unset FOCR_NO_STATS_APPEND; fi
  (cd /tmp && TMPDIR=/tmp rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/tmp_fi_$which -o p -- python3 $repo/bench.py --steps 60 --no-cpu-baseline --no-e2e --no-extra-legs > $out/fetch_inflight_$which.log 2>&1)
  f=$(find $out/tmp_fi_$which -name "*counter_collection.csv" | head -1)
  python3 - "$f" $which <<'PY'
import csv, sys
v = [float(x["Counter_Value"]) for x in csv.DictReader(open(sys.argv[1])) if "scan_mfma2s" in x["Kernel_Name"] and x["Counter_Name"] == "FETCH_SIZE"]
v = v[len(v)//3:]
print(sys.argv[2], "scan kernel FETCH_SIZE KB per launch in the pipeline", round(sum(v) / len(v), 1), "launches", len(v))
PY
  rm -rf $out/tmp_fi_$which
done
