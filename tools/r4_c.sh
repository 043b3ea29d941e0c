repo=$PWD; out=$PWD/gpurun_out/r04; mkdir -p $out
for i in 1 2 3 4 5 6; do
FOCR_PIPE_TRACE=1 FOCR_BENCH_TRACE=1 python3 bench.py --no-cpu-baseline --no-e2e --in-flight 4 --force-gather --steps 60 > $out/g4r$i.json 2> $out/g4r$i.err; echo "[r4_c] g4 run $i: $(cut -c60-110 $out/g4r$i.json)"; grep "\[trace\]" $out/g4r$i.err
done
for i in 1 2 3; do
FOCR_PIPE_TRACE=1 python3 bench.py --no-cpu-baseline --no-e2e --in-flight 4 --force-gather --steps 60 > $out/g4p$i.json 2> $out/g4p$i.err; echo "[r4_c] g4 (pipe trace only) run $i: $(cut -c60-110 $out/g4p$i.json)"
done
