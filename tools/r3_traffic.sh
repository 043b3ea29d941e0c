# only the two PMC passes behind profiles/r03_traffic.json (after an edit to the scan kernel's sources: the profile carries their
# hash); then python tools/r3_collect.py --traffic-only
repo=$PWD; out=$PWD/gpurun_out/r03; mkdir -p $out
for pair in "fetch FETCH_SIZE" "write WRITE_SIZE"; do
  set -- $pair
  (cd /tmp && TMPDIR=/tmp rocprofv3 --pmc $2 --output-format csv -d $out/tmp_$1 -o p -- python3 $repo/tools/kbench.py > $out/$1_pmc_run.log 2>&1)
  f=$(find $out/tmp_$1 -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp $f $out/$1_pmc.csv; rm -rf $out/tmp_$1
  echo "[r3_traffic] $1 done"
done
