#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python bench.py 2> gpurun_out/r5_line.err | tail -1 > gpurun_out/r5_line.json
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r5_line.json").read())
print(round(d["value"]/1e6,3),"M/s ms",round(d["ms_per_step"],2),"grp",round(d["roofline"]["avg_launch_us"],1),"frac",round(d["roofline"]["frac"],3),"exec/alg",round(d["roofline"]["executed_over_algorithmic_flops"],4), d["roofline"]["executed_flops"])
s=d["secondary"]; print("secondary", round(s["value"]/1e6,3),"M/s ms",round(s["ms_per_step"],2),"grp",round(s["roofline"]["avg_launch_us"],1),"frac",round(s["roofline"]["frac"],3), s["roofline"]["traffic_note"])
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
PY
tail -3 gpurun_out/r5_line.err
