#!/bin/bash
# round 5: the register-form statistics kernel: rows per wave (FOCR_S8_ROWS) alone (kprof) and in the pipeline
mkdir -p gpurun_out/r05; o=gpurun_out/r05
for r in 16 24; do
  bash tools/kprof.sh s8_$r KB_POST=1 FOCR_S8_ROWS=$r 2>/dev/null | grep -E "stats|compact_live|scan_mfma2s" | sed "s/^/rows $r: /"
done
FOCR_NO_STATS8=1 bash tools/kprof.sh s8_old KB_POST=1 FOCR_NO_STATS8=1 2>/dev/null | grep -E "stats|scan_mfma2s" | sed "s/^/tiled: /"
for i in 1 2 3; do
  for r in 16 24; do
    FOCR_S8_ROWS=$r python bench.py --steps 300 --no-cpu-baseline --no-e2e --no-extra-legs > $o/s8_$r.json 2>/dev/null
    python -c "import json;d=json.load(open('$o/s8_$r.json'));print('rows $r: value', d['value'], 'ms', d['ms_per_step'], 'scan', d['roofline']['avg_kernel_ms'], 'stats phase', d['phases_ms_per_step']['stats'])"
  done
done
