repo=$PWD; out=$PWD/gpurun_out/r04; mkdir -p $out
bash tools/kprof.sh r04_c3_alone KB_CONFIG=c3 > $out/kprof_c3_alone.log 2>&1; echo "[r4_b] c3 alone done"; tail -30 $out/kprof_c3_alone.log
FOCR_BENCH_TRACE=1 python3 bench.py --no-cpu-baseline --no-e2e --in-flight 4 --force-gather --steps 60 > $out/g4t.json 2> $out/g4t.err; echo "[r4_b] g4 60: $(cut -c1-110 $out/g4t.json)"; grep "\[trace\]" $out/g4t.err
FOCR_BENCH_TRACE=1 python3 bench.py --no-cpu-baseline --no-e2e --in-flight 4 --force-gather --steps 300 > $out/g4t300.json 2> $out/g4t300.err; echo "[r4_b] g4 300: $(cut -c1-110 $out/g4t300.json)"; grep "\[trace\]" $out/g4t300.err
FOCR_BENCH_TRACE=1 python3 bench.py --no-cpu-baseline --no-e2e --in-flight 3 --force-gather --steps 300 > $out/g3t300.json 2> $out/g3t300.err; echo "[r4_b] g3 300: $(cut -c1-110 $out/g3t300.json)"; grep "\[trace\]" $out/g3t300.err
