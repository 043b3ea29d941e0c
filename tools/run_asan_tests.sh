#!/bin/bash
# Non-GPU test suite against the AddressSanitizer + UBSan builds of the CPU-side code (host library, rasteriser, oracle).
# The HIP library is not sanitised (GPU ASan is not available on this pool); tests that only exercise it are left out.
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
make -s -C "$ROOT/font_ocr_amd/csrc" asan
make -s -C "$ROOT/oracle" asan
export FOCR_HOST_LIB_DIR="$ROOT/font_ocr_amd/lib/asan" FOCR_ORACLE_LIB="$ROOT/oracle/asan/liboracle.so"
export LD_PRELOAD="$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so)"
export ASAN_OPTIONS="detect_leaks=0:abort_on_error=1" UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1"
cd "$ROOT"
exec python -m pytest tests -x -q -m "not gpu" -p no:cacheprovider "$@"
