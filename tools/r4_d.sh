repo=$PWD; out=$PWD/gpurun_out/r04; mkdir -p $out
show() { python3 -c "import json,sys; d=json.load(open(sys.argv[1])); print(sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'], d.get('host_gc'))" $1; }
for i in 1 2; do
FOCR_BENCH_GC=observe python3 bench.py --no-cpu-baseline --no-e2e --in-flight 4 --force-gather --steps 60 > $out/gc_obs4_$i.json 2> $out/gc_obs4_$i.err; show $out/gc_obs4_$i.json
python3 bench.py --no-cpu-baseline --no-e2e --in-flight 4 --force-gather --steps 60 > $out/gc_frz4_$i.json 2> $out/gc_frz4_$i.err; show $out/gc_frz4_$i.json
done
FOCR_BENCH_GC=observe python3 bench.py --no-cpu-baseline --no-e2e --in-flight 3 --force-gather --steps 60 > $out/gc_obs3.json 2> $out/gc_obs3.err; show $out/gc_obs3.json
python3 bench.py --no-cpu-baseline --no-e2e --in-flight 3 --force-gather --steps 60 > $out/gc_frz3.json 2> $out/gc_frz3.err; show $out/gc_frz3.json
FOCR_BENCH_GC=observe python3 bench.py --no-cpu-baseline --no-e2e --steps 300 > $out/gc_obs_default.json 2> $out/gc_obs_default.err; show $out/gc_obs_default.json
python3 bench.py --no-cpu-baseline --no-e2e --steps 300 > $out/gc_frz_default.json 2> $out/gc_frz_default.err; show $out/gc_frz_default.json
python3 bench.py --no-cpu-baseline --no-e2e --steps 300 --in-flight 4 > $out/gc_frz_default4.json 2> $out/gc_frz_default4.err; show $out/gc_frz_default4.json
