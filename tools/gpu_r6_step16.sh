#!/bin/bash
# round 6: the 16-row small-minibatch step - parity A/B tests, interleaved bench A/B at one rank's share of cfg3, kernel trace
set -u
REPO=$PWD; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_r6.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -15
EXTRA="--shard-of 8" ROUNDS=${ROUNDS:-2} OUT=$OUT/r6_ab_step16.jsonl bash tools/gpu_exp.sh "cfg3 layerwise CATPPO_STEP16=0" "cfg3 step16" 2>&1 | tee $OUT/r6_ab_step16.txt
rm -rf /tmp/prof_t
(cd /tmp && timeout -s KILL 240 rocprofv3 --kernel-trace -d /tmp/prof_t -- python $REPO/bench.py --workload cfg3 --shard-of 8 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/r6_trace_cfg3_w8.log 2>&1)
DB=$(find /tmp/prof_t -name '*.db' | head -1)
[ -n "$DB" ] && python tools/rocpd_stats.py "$DB" > $OUT/r6_bench_cfg3_shard_kernel_stats.csv
head -14 $OUT/r6_bench_cfg3_shard_kernel_stats.csv | cut -c1-200
