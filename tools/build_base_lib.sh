# builds tools/bin/libfocr_hip_base.so from the HIP sources of another commit (default HEAD), for tools/r4_ab.sh: usage: bash tools/build_base_lib.sh [commit]
set -e
commit=${1:-HEAD}; tmp=$(mktemp -d); mkdir -p tools/bin
git archive $commit font_ocr_amd/csrc/hip include | tar -x -C $tmp
/opt/rocm/bin/hipcc -std=c++17 -O3 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-parameter -I$tmp/include -shared -o tools/bin/libfocr_hip_base.so $tmp/font_ocr_amd/csrc/hip/*.hip
rm -rf $tmp; ls -la tools/bin/libfocr_hip_base.so
