# round 4 baseline data: C3 timeline, the H2D-pipelined leg traced (kernels + copies), the 4-in-flight + gather stamps
repo=$PWD; out=$PWD/gpurun_out/r04; mkdir -p $out
say() { echo "[r4_base] $*"; }
trace() {  # trace <tag> <program args...>: kernel + memcpy trace -> $out/<tag>_kernel_trace.csv, _memory_copy_trace.csv
  tag=$1; shift
  (cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $out/tmp_$tag -o t -- "$@" > $out/${tag}_run.json 2> $out/${tag}.err)
  for k in kernel_trace memory_copy_trace; do f=$(find $out/tmp_$tag -name "*${k}.csv" | head -1); [ -n "$f" ] && cp $f $out/${tag}_${k}.csv; done; rm -rf $out/tmp_$tag
  say "$tag traced: $(cat $out/${tag}_run.json | tail -1)"
}
python3 tools/e2e_leg.py --steps 90 > $out/e2e_plain.json 2>$out/e2e_plain.err; say "e2e un-profiled: $(cat $out/e2e_plain.json)"
python3 tools/e2e_leg.py --steps 90 --resident > $out/res_plain.json 2>$out/res_plain.err; say "resident un-profiled: $(cat $out/res_plain.json)"
python3 tools/e2e_leg.py --steps 90 --in-flight 4 > $out/e2e4_plain.json 2>$out/e2e4_plain.err; say "e2e 4 lanes un-profiled: $(cat $out/e2e4_plain.json)"
trace e2e python3 $repo/tools/e2e_leg.py --steps 90
python3 tools/e2e_timeline.py $out/e2e_kernel_trace.csv $out/e2e_memory_copy_trace.csv > $out/e2e_timeline.log 2>&1
trace res python3 $repo/tools/e2e_leg.py --steps 90 --resident
python3 tools/e2e_timeline.py $out/res_kernel_trace.csv $out/res_memory_copy_trace.csv > $out/res_timeline.log 2>&1
trace c3 python3 $repo/bench.py --no-cpu-baseline --no-e2e --config c3 --pages-per-gpu 64 --steps 24 --warmup 3 --settle-s 0
python3 tools/e2e_timeline.py $out/c3_kernel_trace.csv > $out/c3_timeline.log 2>&1
rm -f $out/*_kernel_trace.csv.gz
FOCR_PIPE_TRACE=1 python3 bench.py --no-cpu-baseline --no-e2e --in-flight 4 --force-gather --steps 60 > $out/g4.json 2> $out/g4.err; say "gather 4: $(cat $out/g4.json | cut -c1-120)"
FOCR_PIPE_TRACE=1 python3 bench.py --no-cpu-baseline --no-e2e --in-flight 3 --force-gather --steps 60 > $out/g3.json 2> $out/g3.err; say "gather 3: $(cat $out/g3.json | cut -c1-120)"
python3 bench.py --no-cpu-baseline --no-e2e --in-flight 4 --steps 60 > $out/n4.json 2> $out/n4.err; say "no gather 4: $(cat $out/n4.json | cut -c1-120)"
# keep the merged-back payload small: the traces are large
for f in $out/*_kernel_trace.csv $out/*_memory_copy_trace.csv; do gzip -f $f; done
ls -la $out
