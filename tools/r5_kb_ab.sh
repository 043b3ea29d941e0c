#!/bin/bash
# round 5: tools/kbench.py (one context, exact sizes: kernels alone on the chip) for the base library and this tree's, alternating
for i in 1 2 3; do
  FOCR_HIP_LIB=$PWD/tools/bin/libfocr_hip_base.so python tools/kbench.py 2>&1 | tail -1 | cut -c1-200 | sed "s/^/base /"
  python tools/kbench.py 2>&1 | tail -1 | cut -c1-200 | sed "s/^/new  /"
done
