# runs the driver's command N times with the lanes' host stamps on; keeps the stamps of the slowest run (gpurun_out/r04/outlier.err)
mkdir -p gpurun_out/r04; best=999999
for i in $(seq 1 ${1:-15}); do
FOCR_PIPE_TRACE=1 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-e2e > gpurun_out/r04/o.json 2> gpurun_out/r04/o.err
v=$(python3 -c "import json; print(int(json.load(open('gpurun_out/r04/o.json'))['value']))")
echo "run $i: $v"
if [ $v -lt $best ]; then best=$v; cp gpurun_out/r04/o.err gpurun_out/r04/outlier.err; cp gpurun_out/r04/o.json gpurun_out/r04/outlier.json; fi
done
echo "slowest: $best"
grep "^\[pipe\]" gpurun_out/r04/outlier.err | tail -24
