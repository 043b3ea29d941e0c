# where do a 20-step timed region's extra milliseconds go?  the lanes' host stamps (FOCR_PIPE_TRACE) of the driver's command
mkdir -p gpurun_out/r04
FOCR_PIPE_TRACE=1 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-e2e > gpurun_out/r04/drain.json 2> gpurun_out/r04/drain.err
python3 -c "
import json; d=json.load(open('gpurun_out/r04/drain.json')); print('value', d['value'], 'ms', d['ms_per_step'])"
grep "^\[pipe\]" gpurun_out/r04/drain.err | tail -32
