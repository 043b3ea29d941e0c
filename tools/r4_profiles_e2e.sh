# round 4, after the early ingest: the three e2e items of tools/r4_profiles.sh again (same outputs under gpurun_out/r04p/)
repo=$PWD; out=$PWD/gpurun_out/r04p; mkdir -p $out
say() { echo "[r4_profiles_e2e] $*"; }
python3 bench.py --steps 300 > $out/bench_c2_default_with_e2e.json 2> $out/bench_c2_default.err; say "default (with the e2e leg and the CPU baseline) done: $(cut -c60-130 $out/bench_c2_default_with_e2e.json)"
(cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $out/tmp_e2e -o t -- python3 $repo/tools/e2e_leg.py --steps 120 > $out/e2e_traced_run.json 2> $out/e2e_traced.err)
for k in kernel_trace memory_copy_trace; do f=$(find $out/tmp_e2e -name "*${k}.csv" | head -1); [ -n "$f" ] && cp $f $out/e2e_${k}.csv; done; rm -rf $out/tmp_e2e
python3 tools/e2e_timeline.py $out/e2e_kernel_trace.csv $out/e2e_memory_copy_trace.csv > $out/e2e_timeline.log 2>&1; rm -f $out/e2e_kernel_trace.csv $out/e2e_memory_copy_trace.csv; say "e2e timeline done"
rm -f $out/e2e_legs.jsonl; for rep in 1 2; do for v in "" "--no-prefetch" "--resident" "--in-flight 4" "--in-flight 4 --resident"; do python3 tools/e2e_leg.py --steps 240 $v >> $out/e2e_legs.jsonl 2>/dev/null; done; done; say "e2e legs:"; cat $out/e2e_legs.jsonl
head -30 $out/e2e_timeline.log
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver_cmd.json 2> $out/bench_driver_cmd.err; say "the driver's own command done: $(cut -c60-130 $out/bench_driver_cmd.json)"
