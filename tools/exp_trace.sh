#!/bin/bash
# kernel trace of the default bench under an environment switch:  bash tools/exp_trace.sh TAG [VAR=value ...]
set -u
REPO=$PWD; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
TAG=$1; shift
rm -rf /tmp/prof_$TAG
(cd /tmp && env "$@" timeout -s KILL 240 rocprofv3 --kernel-trace -d /tmp/prof_$TAG -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/trace_$TAG.log" 2>&1)
DB=$(find /tmp/prof_$TAG -name '*.db' | head -1)
[ -n "$DB" ] && python tools/rocpd_stats.py "$DB" > "$OUT/kernel_stats_$TAG.csv"
python - "$OUT/kernel_stats_$TAG.csv" <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:12]:
    if 'gae_scan' in r['kernel'] or 'at::native' in r['kernel']: continue
    print("%-60s %10s calls %4s avg %8s"%(r['kernel'][:60],r['blocks'],r['calls'],r['avg_us']))
PY
