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
timeout 900 python -m pytest tests/test_gpu_r2_features.py -m gpu -q --timeout 600 -p no:cacheprovider 2>&1 | tail -3 | cut -c1-200
