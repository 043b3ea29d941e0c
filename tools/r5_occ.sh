#!/bin/bash
# round 5: the statistics kernel forced to eight waves per SIMD (this tree) against tools/bin/libfocr_hip_base.so: alone and in the pipeline
mkdir -p gpurun_out/r05
for which in base new; do
  if [ $which = base ]; then L="FOCR_HIP_LIB=$PWD/tools/bin/libfocr_hip_base.so"; else L="FOCR_X=1"; fi
  bash tools/kprof.sh occ KB_POST=1 $L 2>/dev/null > gpurun_out/occ.txt; grep stats8 gpurun_out/occ.txt | sed "s/^/$which: /"
done
bash tools/r5_ab_phases.sh
