#!/bin/bash
# round 5: shape-generic row-resident forward on the reference's own network (512 / 256 / 128): interleaved A/B + a kernel trace
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
ROUNDS=${ROUNDS:-3} OUT=gpurun_out/r5_ab_rows_wide.jsonl bash tools/gpu_exp.sh "reference layerwise CATPPO_ROWS_WIDE=0" "reference rows_wide CATPPO_ROWS_WIDE=1" "reference wide_train_only CATPPO_ROWS_WIDE_ROLLOUT=0" > gpurun_out/r5_ab_rows_wide.txt 2>&1
tail -5 gpurun_out/r5_ab_rows_wide.txt
bash tools/gpu_trace_one.sh reference r5 > gpurun_out/r5_trace_reference.txt 2>&1
python - <<'PY'
import csv
for r in list(csv.DictReader(open("gpurun_out/r5_bench_reference_kernel_stats.csv")))[:14]:
    k=r["kernel"].replace("(anonymous namespace)::","").replace("void ","").split("(")[0][:80]
    print("  %-82s %-10s calls %5s avg %8s vgpr %s lds %s"%(k,r["blocks"],r["calls"],r["avg_us"],r["vgpr"],r["lds_bytes"]))
PY
