# A/B of scan-kernel builds on one box (tools/kbench.py: every kernel alone on the chip, exact sizes): a library built from another
# commit (tools/bin/libfocr_hip_base.so) against this tree's, alternating
set -o pipefail
mkdir -p gpurun_out
for rep in 1 2 3; do
  echo "base:"; FOCR_HIP_LIB=$PWD/tools/bin/libfocr_hip_base.so python tools/kbench.py 2>&1 | tail -1
  echo "new:"; python tools/kbench.py 2>&1 | tail -1
done
