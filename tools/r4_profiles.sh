# round 4: every profile behind profiles/r04_* (run on the GPU box from the repo root; outputs under gpurun_out/r04p/)
repo=$PWD; out=$PWD/gpurun_out/r04p; mkdir -p $out
say() { echo "[r4_profiles] $*"; }
stats() {  # stats <tag> <program args...>: rocprofv3 kernel stats of a run -> $out/<tag>_kernel_stats.csv, its stdout -> $out/<tag>_run.json
  tag=$1; shift
  (cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d $out/tmp_$tag -o k -- "$@" > $out/${tag}_run.json 2> $out/${tag}.err)
  f=$(find $out/tmp_$tag -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $out/${tag}_kernel_stats.csv; rm -rf $out/tmp_$tag
  say "$tag done: $(cut -c60-130 $out/${tag}_run.json)"
}
pmc() {  # pmc <tag> "<counters>" <program args...> -> $out/<tag>_pmc.csv
  tag=$1; ctr=$2; shift 2
  (cd /tmp && TMPDIR=/tmp rocprofv3 --pmc $ctr --output-format csv -d $out/tmp_$tag -o p -- "$@" > $out/${tag}_pmc_run.log 2>&1)
  f=$(find $out/tmp_$tag -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp $f $out/${tag}_pmc.csv; rm -rf $out/tmp_$tag
  say "$tag pmc done"
}
B="python3 $repo/bench.py --no-cpu-baseline --no-e2e --steps 100"
stats bench_c2 $B
stats bench_c2_serial $B --in-flight 1 --steps 40
stats bench_c2_noise $B --noise
stats bench_c3 $B --config c3 --pages-per-gpu 64 --steps 12 --warmup 2
python3 bench.py --no-cpu-baseline --no-e2e --config c3 --pages-per-gpu 64 --c3-pages 1024 --steps 2 --warmup 1 > $out/bench_c3_stream_1024pages.json 2> $out/bench_c3_stream.err; say "c3 stream done"
python3 bench.py --no-cpu-baseline --no-e2e --config c4 --c4-pages 2048 --steps 3 --warmup 1 > $out/bench_c4_2048pages_1gpu.json 2> $out/bench_c4.err; say "c4 done"
python3 bench.py --no-cpu-baseline --steps 300 > $out/bench_c2_default_with_e2e.json 2> $out/bench_c2_default.err; say "default (with the e2e leg) done"
python3 bench.py --no-cpu-baseline --no-e2e --force-gather --steps 300 > $out/bench_c2_force_gather.json 2> $out/bench_c2_force_gather.err; say "force-gather done"
python3 bench.py --no-cpu-baseline --no-e2e --force-gather --in-flight 4 --steps 300 > $out/bench_c2_force_gather_inflight4.json 2> /dev/null; say "force-gather 4 done"
python3 bench.py --no-cpu-baseline --no-e2e --in-flight 4 --steps 300 > $out/bench_c2_inflight4.json 2> /dev/null; say "in-flight 4 done"
python3 bench.py --no-cpu-baseline --no-e2e --steps 300 > $out/bench_c2_300.json 2> /dev/null; say "in-flight 3, 300 steps done"
bash tools/timeline.sh > $out/timeline_bench_c2.log 2>&1; say "timeline done"
# the PCIe-inclusive leg, kernels + copies traced
(cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $out/tmp_e2e -o t -- python3 $repo/tools/e2e_leg.py --steps 120 > $out/e2e_traced_run.json 2> $out/e2e_traced.err)
for k in kernel_trace memory_copy_trace; do f=$(find $out/tmp_e2e -name "*${k}.csv" | head -1); [ -n "$f" ] && cp $f $out/e2e_${k}.csv; done; rm -rf $out/tmp_e2e
python3 tools/e2e_timeline.py $out/e2e_kernel_trace.csv $out/e2e_memory_copy_trace.csv > $out/e2e_timeline.log 2>&1; rm -f $out/e2e_kernel_trace.csv $out/e2e_memory_copy_trace.csv; say "e2e timeline done"
rm -f $out/e2e_legs.jsonl $out/bench_thr_sweep.log; for v in "" "--no-prefetch" "--resident"; do python3 tools/e2e_leg.py --steps 200 $v >> $out/e2e_legs.jsonl 2>/dev/null; done; say "e2e legs: $(cat $out/e2e_legs.jsonl | tr '\n' ' ')"
# threshold sweep: value, scan ms, candidates, hits, redone batches
for thr in 0.5 0.8 0.9 0.97; do
python3 bench.py --no-cpu-baseline --no-e2e --steps 100 --threshold $thr 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; p=d['phases_ms_per_step']
print('thr', $thr, 'value', d['value'], 'ms_per_step', d['ms_per_step'], 'scan launch ms in flight', r['avg_kernel_ms'], 'alone', r.get('isolated_avg_kernel_ms'), 'frac', r['frac'], 'frac_issued', r['frac_issued'], 'lane phases ms (stats, scan, tail, order, post):', p['stats'], p['scan'], p['verify'], p['order'], p['process_hits'], 'work', d['work'], 'redone', d['size_estimates']['batches_redone_exact'])" >> $out/bench_thr_sweep.log
done; say "thr sweep done"; cat $out/bench_thr_sweep.log
pmc fetch "FETCH_SIZE" python3 $repo/tools/kbench.py
pmc write "WRITE_SIZE" python3 $repo/tools/kbench.py
pmc mfma "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" python3 $repo/tools/kbench.py
pmc insts "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_ANY" python3 $repo/tools/kbench.py
bash tools/pmc_cu_busy.sh r04p_c2 > $out/cu_busy_c2.log 2>&1; bash tools/pmc_cu_busy.sh r04p_c3 KB_CONFIG=c3 > $out/cu_busy_c3.log 2>&1; say "cu busy done"
bash tools/kprof.sh r04p_c2_alone > $out/kprof_c2_alone.log 2>&1; bash tools/kprof.sh r04p_c3_alone KB_CONFIG=c3 > $out/kprof_c3_alone.log 2>&1; say "kprof done"
ls $out | head -80
