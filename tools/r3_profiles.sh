# round 3: every profile behind profiles/r03_* (run on the GPU box from the repo root; outputs under gpurun_out/r03/)
repo=$PWD; out=$PWD/gpurun_out/r03; mkdir -p $out
say() { echo "[r3_profiles] $*"; }
stats() {  # stats <tag> <program args...>: rocprofv3 kernel stats of a run -> $out/<tag>_kernel_stats.csv, its stdout -> $out/<tag>_run.json
  tag=$1; shift
  (cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d $out/tmp_$tag -o k -- "$@" > $out/${tag}_run.json 2> $out/${tag}.err)
  f=$(find $out/tmp_$tag -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $out/${tag}_kernel_stats.csv; rm -rf $out/tmp_$tag
  say "$tag done"
}
pmc() {  # pmc <tag> "<counters>" <program args...> -> $out/<tag>_pmc.csv
  tag=$1; ctr=$2; shift 2
  (cd /tmp && TMPDIR=/tmp rocprofv3 --pmc $ctr --output-format csv -d $out/tmp_$tag -o p -- "$@" > $out/${tag}_pmc_run.log 2>&1)
  f=$(find $out/tmp_$tag -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp $f $out/${tag}_pmc.csv; rm -rf $out/tmp_$tag
  say "$tag pmc done"
}
B="python3 $repo/bench.py --no-cpu-baseline --no-e2e --steps 100"  # the committed set was taken at 100 timed steps (the default is 300 now)
stats bench_c2 $B
stats bench_c2_serial $B --in-flight 1 --steps 40
stats bench_c2_noise $B --noise
stats bench_c3 $B --config c3 --pages-per-gpu 64 --steps 12 --warmup 2
python3 bench.py --no-cpu-baseline --no-e2e --config c4 --c4-pages 2048 --steps 3 --warmup 1 > $out/bench_c4_2048pages_1gpu.json 2> $out/bench_c4.err; say "c4 done"
python3 bench.py --no-cpu-baseline --with-upload --steps 60 > $out/bench_c2_upload.json 2> $out/bench_c2_upload.err; say "upload done"
python3 bench.py --no-cpu-baseline --no-e2e --force-gather --steps 60 > $out/bench_c2_force_gather.json 2> $out/bench_c2_force_gather.err; say "force-gather done"
python3 bench.py --no-cpu-baseline --no-e2e --in-flight 4 --steps 60 > $out/bench_c2_inflight4.json 2> /dev/null; say "in-flight 4 done"
bash tools/timeline.sh > $out/timeline_bench_c2.log 2>&1; say "timeline done"
tools/bin/mfma_shape > $out/mfma_shape.log 2>&1; say "mfma_shape done"
pmc fetch "FETCH_SIZE" python3 $repo/tools/kbench.py
pmc write "WRITE_SIZE" python3 $repo/tools/kbench.py
pmc mfma "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" python3 $repo/tools/kbench.py
pmc insts "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_ANY" python3 $repo/tools/kbench.py
# BASELINE configs[4]: product kernels (i8 MFMA prefilter vs v_dot4 direct) and the same loop in i8 / bf16 MFMA form
python3 tools/c5_gemm_variant.py > $out/c5_product.json 2> $out/c5_product.err; say "c5 product done"
pmc c5_product "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" python3 $repo/tools/c5_gemm_variant.py
$repo/tools/bin/c5_forms 64 > $out/c5_forms.json 2> $out/c5_forms.err; say "c5 forms done: $(cat $out/c5_forms.json)"
pmc c5_forms "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" $repo/tools/bin/c5_forms 64
ls -la $out
