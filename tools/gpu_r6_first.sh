#!/bin/bash
# round 6, first call: MFMA chain-order probe + baseline lines of the round-5 tree on this round's box
set -u
mkdir -p gpurun_out
./tools/bin/mfma16_probe > gpurun_out/r6_mfma16_probe.txt 2>&1; cat gpurun_out/r6_mfma16_probe.txt
B="timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline"
$B --workload cfg3 --shard-of 8 2>/dev/null | tail -1 > gpurun_out/r6_base_cfg3_w8.json
$B --workload cfg2 2>/dev/null | tail -1 > gpurun_out/r6_base_cfg2.json
$B --workload reference 2>/dev/null | tail -1 > gpurun_out/r6_base_reference.json
python - <<'PY'
import json
for f in ("cfg3_w8","cfg2","reference"):
    d=json.loads(open(f"gpurun_out/r6_base_{f}.json").read())
    print(f, round(d['value']/1e6,3), "M/s ms", round(d['ms_per_step'],3), "grp_us", round(d['roofline']['avg_launch_us'],1), d['phases_device_ms'])
PY
