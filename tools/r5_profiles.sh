# round 5: every profile behind profiles/r05_* (run on the GPU box from the repo root; outputs under gpurun_out/r05p/)
repo=$PWD; out=$PWD/gpurun_out/r05p; mkdir -p $out
say() { echo "[r5_profiles] $*"; }
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
B="python3 $repo/bench.py --no-cpu-baseline --no-e2e --no-extra-legs --steps 100"
stats bench_c2 $B
stats bench_c2_serial $B --in-flight 1 --depth 1 --steps 40
stats bench_c2_noise $B --noise
stats bench_c3 $B --config c3 --pages-per-gpu 64 --steps 12 --warmup 2
python3 bench.py --steps 300 > $out/bench_c2_default_300steps.json 2> $out/bench_c2_default.err; say "default, 300 steps, every leg: done"
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver_cmd_20steps.json 2> $out/bench_driver_cmd.err; say "the driver's command: done"
python3 bench.py --no-cpu-baseline --no-e2e --no-extra-legs --force-gather --steps 300 > $out/bench_c2_force_gather.json 2> $out/bench_c2_force_gather.err; say "force-gather done"
python3 bench.py --no-cpu-baseline --no-e2e --no-extra-legs --steps 300 --inject-stall-ms 3 > $out/bench_c2_stall3.json 2> /dev/null; say "3 ms stalls done"
python3 bench.py --no-cpu-baseline --no-e2e --no-extra-legs --steps 300 --inject-stall-ms 6 > $out/bench_c2_stall6.json 2> /dev/null; say "6 ms stalls done"
python3 bench.py --no-cpu-baseline --no-e2e --no-extra-legs --steps 300 --inject-stall-ms 3 --depth 1 > $out/bench_c2_stall3_depth1.json 2> /dev/null; say "3 ms stalls, depth 1 done"
python3 bench.py --no-cpu-baseline --no-e2e --no-extra-legs --steps 300 --depth 1 > $out/bench_c2_depth1.json 2> /dev/null; say "depth 1 done"
python3 bench.py --no-cpu-baseline --no-e2e --no-extra-legs --config c4 --c4-pages 8192 --steps 1 --warmup 1 > $out/bench_c4_8192pages_1gpu.json 2> $out/bench_c4.err; say "c4 (8192 pages on one rank) done"
python3 bench.py --no-cpu-baseline --no-e2e --no-extra-legs --config c3 --pages-per-gpu 64 --c3-pages 1024 --steps 2 --warmup 1 > $out/bench_c3_stream_1024pages.json 2> $out/bench_c3_stream.err; say "c3 stream done"
bash tools/timeline.sh > $out/timeline_bench_c2.log 2>&1; say "timeline done"
for thr in 0.5 0.8 0.9 0.97; do
python3 bench.py --no-cpu-baseline --no-e2e --no-extra-legs --steps 100 --threshold $thr 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; p=d['phases_ms_per_step']
print('thr', $thr, 'value', d['value'], 'ms_per_step', d['ms_per_step'], 'scan launch ms in flight', r['avg_kernel_ms'], 'alone', r.get('isolated_avg_kernel_ms'), 'frac', r['frac'], 'frac_issued', r['frac_issued'], 'lane phases ms (stats, scan, tail, order, post):', p['stats'], p['scan'], p['verify'], p['order'], p['process_hits'], 'work', d['work'], 'redone', d['size_estimates']['batches_redone_exact'])" >> $out/bench_thr_sweep.log
done; say "thr sweep done"; cat $out/bench_thr_sweep.log
pmc fetch "FETCH_SIZE" python3 $repo/tools/kbench.py
pmc write "WRITE_SIZE" python3 $repo/tools/kbench.py
pmc mfma "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" python3 $repo/tools/kbench.py
pmc insts "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_ANY" python3 $repo/tools/kbench.py
bash tools/pmc_cu_busy.sh r05p_c2 KB_POST=1 > $out/cu_busy_c2.log 2>&1; bash tools/pmc_cu_busy.sh r05p_c3 KB_CONFIG=c3 KB_POST=1 > $out/cu_busy_c3.log 2>&1; say "cu busy done"
bash tools/kprof.sh r05p_c2_alone KB_POST=1 > $out/kprof_c2_alone.log 2>&1; bash tools/kprof.sh r05p_c3_alone KB_CONFIG=c3 KB_POST=1 > $out/kprof_c3_alone.log 2>&1; say "kprof done"
python3 tools/cli_e2e.py --pages 4096 > $out/cli_e2e_4096_pgm.json 2> $out/cli_e2e.err; say "cli e2e: $(cut -c1-200 $out/cli_e2e_4096_pgm.json)"
ls $out | head -80
