repo=$PWD; out=$PWD/gpurun_out/r04; mkdir -p $out
for kb in 80 150 110; do
FOCR_VERIFY_CHUNK_KB=$kb bash tools/kprof.sh r04_j_c3_kb$kb KB_CONFIG=c3 FOCR_VERIFY_CHUNK_KB=$kb > $out/j_kprof_c3_kb$kb.log 2>&1; echo "chunk budget $kb KB:"; grep -E "verify|scan_mfma" $out/j_kprof_c3_kb$kb.log
done
show() { python3 -c "import json,sys; d=json.load(open(sys.argv[1])); r=d['roofline']; print(sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'], r['avg_kernel_ms'])" $1; }
for kb in 80 150 80 150; do FOCR_VERIFY_CHUNK_KB=$kb python3 bench.py --no-cpu-baseline --no-e2e --config c3 --pages-per-gpu 64 --steps 12 --warmup 2 > $out/j_c3_kb$kb.json 2> $out/j_c3_kb$kb.err; show $out/j_c3_kb$kb.json; done
