# MFMA-pipe and clock counters of the scan kernel (run on the GPU box from the repo root; outputs under gpurun_out/pmc_mfma/)
out=$PWD/gpurun_out/pmc_mfma; repo=$PWD; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for set in "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU" "SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $out/p$i -o p -- python3 $repo/tools/kbench.py > $out/p$i.log 2>&1 || echo "pass $i failed"
  f=$(find $out/p$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && cp $f $out/pass$i.csv
  rm -rf $out/p$i
done
cd $repo
python3 - <<PY
import csv, glob
for f in sorted(glob.glob("gpurun_out/pmc_mfma/pass*.csv")):
    by = {}
    for x in csv.DictReader(open(f)):
        if "scan_mfma2s" in x["Kernel_Name"]:
            by.setdefault(x["Counter_Name"], []).append(float(x["Counter_Value"]))
    for k, v in by.items(): print(f.split("/")[-1], k, len(v), sum(v) / len(v))
PY
