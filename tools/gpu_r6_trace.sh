#!/bin/bash
# kernel trace of one workload: bash tools/gpu_r6_trace.sh <workload> <suffix> [extra bench args]
set -u
REPO=$PWD; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
WL=$1; SFX=$2; shift 2
rm -rf /tmp/prof_t
(cd /tmp && timeout -s KILL 240 rocprofv3 --kernel-trace -d /tmp/prof_t -- python $REPO/bench.py --workload $WL --steps 3 --warmup 1 --no-cpu-baseline "$@" > $OUT/r6_trace_$SFX.log 2>&1)
DB=$(find /tmp/prof_t -name '*.db' | head -1)
[ -n "$DB" ] && python tools/rocpd_stats.py "$DB" > $OUT/r6_bench_${SFX}_kernel_stats.csv
head -12 $OUT/r6_bench_${SFX}_kernel_stats.csv | cut -c1-180
