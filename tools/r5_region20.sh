#!/bin/bash
# round 5: where a 20-step timed region's time goes (the driver's command with the per-ticket phase stamps)
mkdir -p gpurun_out/r05
for i in 1 2 3; do
  FOCR_BENCH_DUMP_TICKETS=gpurun_out/r05/d20_$i.json python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-e2e --no-extra-legs > gpurun_out/r05/d20_run_$i.json 2>/dev/null
  python - <<PY
import json
t=json.load(open("gpurun_out/r05/d20_$i.json")); d=json.load(open("gpurun_out/r05/d20_run_$i.json"))
t0=t[0]["stats_start"]
print("run $i value", d["value"], "ms/step", d["ms_per_step"], "region ms", round(d["ms_per_step"]*20,3), "| device: first stats start -> last post end", round(t[-1]["post_end"]-t0,3), "| first scan start", round(t[0]["scan_launch_start"]-t0,3), "| scans", round(sum(x["scan_launch_end"]-x["scan_launch_start"] for x in t),3), "| gaps", round(sum(b["scan_launch_start"]-a["scan_launch_end"] for a,b in zip(t,t[1:])),3), "| last tail", round(t[-1]["post_end"]-t[-1]["scan_launch_end"],3), "| first completion", d["step_stats"]["first_completion_ms"])
PY
done
