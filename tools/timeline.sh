# kernel timeline of the default bench (three batches in flight): rocprofv3 --kernel-trace of 150 timed steps (no settle loop: its fence every
# six steps would put a drained pipeline into every other round of the trace) -> gpurun_out/timeline_kernel_trace.csv,
# then tools/timeline.py: what the GPU does between one scan launch and the next (run on the GPU box from the repo root)
repo=$PWD; out=$PWD/gpurun_out; mkdir -p $out
(cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --output-format csv -d $out/tmp_timeline -o t -- python3 $repo/bench.py --no-cpu-baseline --no-e2e --no-extra-legs --steps 150 --warmup 9 --settle-s 0 "$@" > $out/timeline_run.json 2> $out/timeline.err)
f=$(find $out/tmp_timeline -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && cp $f $out/timeline_kernel_trace.csv; rm -rf $out/tmp_timeline
python3 tools/timeline.py $out/timeline_kernel_trace.csv
