# PMC passes over tools/kbench.py for the tail's kernels (run on the GPU box from the repo root)
repo=$PWD; out=$PWD/gpurun_out/pmc_tail; mkdir -p $out
i=0
for set in "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  (cd /tmp && TMPDIR=/tmp rocprofv3 --pmc $set --output-format csv -d $out/p$i -o p -- python3 $repo/tools/kbench.py > $out/p$i.log 2>&1) || echo "pass $i failed"
  f=$(find $out/p$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && cp $f $out/pass$i.csv
  rm -rf $out/p$i
done
python3 - <<'PY'
import csv, glob, collections
names = ["verify_flat", "stats_kernel", "row_sort_kernel<1024", "row_scatter", "unit_emit", "scan_mfma2s"]
for f in sorted(glob.glob("gpurun_out/pmc_tail/pass*.csv")):
    by = collections.defaultdict(lambda: collections.defaultdict(list))
    for x in csv.DictReader(open(f)):
        for n in names:
            if n in x["Kernel_Name"]:
                by[n][x["Counter_Name"]].append(float(x["Counter_Value"]))
                by[n]["_dur_us"].append((int(x["End_Timestamp"]) - int(x["Start_Timestamp"])) / 1e3)
    for n, c in by.items():
        print(f.split("/")[-1], n, {k: round(sum(v) / len(v)) for k, v in c.items()})
PY
