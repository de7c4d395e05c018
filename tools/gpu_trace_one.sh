#!/bin/bash
# kernel trace of one bench run only (no counter passes):  bash tools/gpu_trace_one.sh [workload] [tag]
set -u
WL=${1:-cfg2}; TAG=${2:-exp}
OUT=$PWD/gpurun_out; mkdir -p "$OUT"; export TMPDIR=/tmp
CMD="python $PWD/bench.py --workload $WL --steps 3 --warmup 1 --no-cpu-baseline ${BENCH_ARGS:-}"
dir=/tmp/prof_trace; rm -rf $dir
(cd /tmp && timeout -s KILL 240 rocprofv3 --kernel-trace -d $dir -- $CMD > "$OUT/${TAG}_trace.log" 2>&1)
DB=$(find $dir -name '*.db' | head -1)
[ -n "$DB" ] && python tools/rocpd_stats.py "$DB" > "$OUT/${TAG}_bench_${WL}_kernel_stats.csv"
head -12 "$OUT/${TAG}_bench_${WL}_kernel_stats.csv" | cut -c1-150
