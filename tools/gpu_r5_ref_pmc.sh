#!/bin/bash
# HBM traffic of the minibatch group at the reference's own network (512 / 256 / 128): FETCH / WRITE passes + kernel trace,
# then a bench line that carries the stamped summary
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out profiles
SQ_PASSES=0 bash tools/profile_bench.sh reference r5 > gpurun_out/ref_profile.log 2>&1
cp gpurun_out/r5_pmc_traffic_reference.json gpurun_out/r5_bench_reference_kernel_stats.csv profiles/ 2>/dev/null
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --workload reference 2>/dev/null | tail -1 > gpurun_out/r5_bench_reference.json
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5_bench_reference.json").read()); r = d["roofline"]
print("reference", round(d["value"]/1e6, 3), round(d["ms_per_step"], 2), round(r["avg_launch_us"], 1), r["frac"], r.get("frac_profiled"), r["traffic"], r.get("hbm_GBps"), r.get("traffic_note"))
print(open("gpurun_out/r5_pmc_traffic_reference.json").read()[:600])
PY
