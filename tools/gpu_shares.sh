#!/bin/bash
# strong-scaling shares of cfg3 (one rank's work at W ranks, single process) + kernel traces of the shard and the reference shapes
set -u
REPO=$PWD; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
B="timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline"
: > "$OUT/r4_bench_cfg3_shares.jsonl"
for W in 1 2 4 8; do $B --workload cfg3 --shard-of $W 2>/dev/null | tail -1 >> "$OUT/r4_bench_cfg3_shares.jsonl"; done
$B --workload cfg3_shard 2>/dev/null | tail -1 > "$OUT/r4_bench_cfg3_shard.json"
for WL in cfg3_shard reference; do
  rm -rf /tmp/prof_t
  (cd /tmp && timeout -s KILL 240 rocprofv3 --kernel-trace -d /tmp/prof_t -- python $REPO/bench.py --workload $WL --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/r4_trace_$WL.log" 2>&1)
  DB=$(find /tmp/prof_t -name '*.db' | head -1)
  [ -n "$DB" ] && python tools/rocpd_stats.py "$DB" > "$OUT/r4_bench_${WL}_kernel_stats.csv"
done
python - <<'PY'
import json
for l in open('gpurun_out/r4_bench_cfg3_shares.jsonl'):
    d=json.loads(l); print(d['config'].get('simulated_shard_of_world'), round(d['value']/1e6,3), round(d['ms_per_step'],2), round(d['roofline']['avg_launch_us'],1), d['config'].get('one_call_optimiser_step'))
d=json.loads(open('gpurun_out/r4_bench_cfg3_shard.json').read()); print('cfg3_shard', round(d['value']/1e6,3), round(d['ms_per_step'],2), round(d['roofline']['avg_launch_us'],1), d['config'].get('one_call_optimiser_step'))
PY
head -14 gpurun_out/r4_bench_cfg3_shard_kernel_stats.csv | cut -c1-150
