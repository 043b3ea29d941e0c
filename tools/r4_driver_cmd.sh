# the driver's own command (BENCH_r03.json: "python3 bench.py --gpus 1 --steps 20 --warmup 5"), three times, then 300 steps once
mkdir -p gpurun_out/r04
for i in 1 2 3; do python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04/driver_cmd_$i.json 2> gpurun_out/r04/driver_cmd_$i.err || exit 1
python3 -c "
import json; d=json.load(open('gpurun_out/r04/driver_cmd_$i.json')); r=d['roofline']
print('20 steps run $i: value', d['value'], 'ms', d['ms_per_step'], 'e2e', d.get('e2e_value_incl_h2d_pipelined'), 'frac', r['frac'], 'issued', r['frac_issued'], 'isolated', r.get('frac_isolated'), 'whole', r['frac_whole_step'], 'cpu', d['cpu_baseline']['value'], d.get('optional_leg_errors'))"; done
python3 bench.py --no-cpu-baseline --no-e2e > gpurun_out/r04/driver_cmd_300.json 2>/dev/null; python3 -c "
import json; d=json.load(open('gpurun_out/r04/driver_cmd_300.json')); print('300 steps: value', d['value'], 'ms', d['ms_per_step'])"
