#!/bin/bash
# strong-scaling shares of cfg3 (one rank's work at W ranks, single process), window A/B of the 16-row step at 4096 / 8192 rows
set -u
OUT=$PWD/gpurun_out; mkdir -p $OUT
B="timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline"
: > "$OUT/r6_bench_cfg3_shares.jsonl"
for W in 1 2 4 8; do $B --workload cfg3 --shard-of $W 2>/dev/null | tail -1 >> "$OUT/r6_bench_cfg3_shares.jsonl"; done
python - <<'PY'
import json
for l in open('gpurun_out/r6_bench_cfg3_shares.jsonl'):
    d=json.loads(l); print("W", d['config'].get('simulated_shard_of_world'), round(d['value']/1e6,3), "M/s  ms", round(d['ms_per_step'],2), "grp_us", round(d['roofline']['avg_launch_us'],1), d['phases_device_ms'])
PY
EXTRA="--shard-of 4" ROUNDS=2 OUT=$OUT/r6_ab_step16_w4.jsonl bash tools/gpu_exp.sh "cfg3 w4_layerwise CATPPO_STEP16=0" "cfg3 w4_step16" 2>&1 | tail -3 | tee $OUT/r6_ab_step16_window.txt
EXTRA="--shard-of 2" ROUNDS=2 OUT=$OUT/r6_ab_step16_w2.jsonl bash tools/gpu_exp.sh "cfg3 w2_default" "cfg3 w2_step16 CATPPO_STEP16_MAX_ROWS=8192" 2>&1 | tail -3 | tee -a $OUT/r6_ab_step16_window.txt
ROUNDS=2 OUT=$OUT/r6_ab_step16_cfg1.jsonl bash tools/gpu_exp.sh "cfg1 layerwise CATPPO_STEP16=0 CATPPO_STEP16_FWD=0" "cfg1 step16" 2>&1 | tail -3 | tee -a $OUT/r6_ab_step16_window.txt
