#!/bin/bash
# kernel traces of the other workloads (evidence for DESIGN.md 5): rocprofv3 --kernel-trace summaries
set -u
REPO=$PWD
OUT=$REPO/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
for WL in reference cfg3_shard cfg4 cfg5; do
  dir=/tmp/prof_$WL; rm -rf $dir
  (cd /tmp && timeout -s KILL 240 rocprofv3 --kernel-trace -d $dir -- python $REPO/bench.py --workload $WL --steps 3 --warmup 1 --no-cpu-baseline --set graph_update=False > "$OUT/trace_$WL.log" 2>&1)
  DB=$(find $dir -name '*.db' | head -1)
  [ -n "$DB" ] && python tools/rocpd_stats.py "$DB" > "$OUT/r2_bench_${WL}_kernel_stats.csv" && echo "$WL ok"
done
# strong-scaling shares of cfg3 (16384 envs, 16384 minibatch GLOBAL): one rank's share at 1 / 2 / 4 / 8 ranks, no collectives
: > "$OUT/r2_bench_cfg3_shares.jsonl"
for W in 1 2 4 8; do
  timeout 300 python bench.py --workload cfg3 --shard-of $W --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 >> "$OUT/r2_bench_cfg3_shares.jsonl"
done
python - "$OUT/r2_bench_cfg3_shares.jsonl" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.strip():
        d = json.loads(l)
        print("cfg3 share of", d["config"]["simulated_shard_of_world"], "ranks:", round(d["value"] / 1e6, 3), "M/s per rank, ms", round(d["ms_per_step"], 2))
PY
