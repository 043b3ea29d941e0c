repo=$PWD; out=$PWD/gpurun_out/r04; mkdir -p $out
show() { python3 -c "import json,sys; d=json.load(open(sys.argv[1])); r=d['roofline']; print(sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'], r['avg_kernel_ms'], d['config'].get('cu_partition'))" $1; }
for q in 16 32; do GPU_MAX_HW_QUEUES=$q python3 bench.py --no-cpu-baseline --no-e2e --steps 100 --cu-split 4 > $out/h_q$q.json 2> $out/h_q$q.err; show $out/h_q$q.json; done
(cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --output-format csv -d $out/tmp_h -o t -- python3 $repo/bench.py --no-cpu-baseline --no-e2e --steps 60 --warmup 9 --settle-s 0 --cu-split 4 > $out/h_trace_run.json 2> $out/h_trace.err)
f=$(find $out/tmp_h -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && cp $f $out/h_kernel_trace.csv; rm -rf $out/tmp_h
python3 tools/e2e_timeline.py $out/h_kernel_trace.csv --list 3 > $out/h_timeline.log 2>&1; head -80 $out/h_timeline.log; gzip -f $out/h_kernel_trace.csv
