# A/B of experiment builds of the HIP library with tools/kbench.py (scan kernel alone, exact sizes)
for v in ${AB_LIBS:-A B A B}; do
FOCR_HIP_LIB=$PWD/font_ocr_amd/lib/exp/libfocr_hip_$v.so python tools/kbench.py 2>&1 | tail -1 | cut -c1-150 | sed "s/^/$v /"
done
