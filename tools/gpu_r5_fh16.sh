#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_end_to_end.py tests/test_gpu_r4.py -m gpu -q -x -p no:cacheprovider -k "minibatch or fused_head or iteration or one_call or row_resident_forward_equals" 2>&1 | tail -3
OLD=$PWD/tools/bin/libcatppo_fh32.so
ROUNDS=3 OUT=gpurun_out/r5_ab_fwd_head_mfma16.jsonl bash tools/gpu_exp.sh "cfg2 mfma32x32 CATPPO_LIB=$OLD" "cfg2 mfma16x16 X=1" "reference mfma32x32 CATPPO_LIB=$OLD" "reference mfma16x16 X=1" > gpurun_out/r5_ab_fwd_head_mfma16.txt 2>&1
tail -5 gpurun_out/r5_ab_fwd_head_mfma16.txt
bash tools/gpu_trace_one.sh cfg2 r5fh > /dev/null 2>&1
python - <<'PY'
import csv
for r in list(csv.DictReader(open("gpurun_out/r5fh_bench_cfg2_kernel_stats.csv")))[:6]:
    k=r["kernel"].replace("(anonymous namespace)::","").replace("void ","").split("(")[0][:74]
    print("  %-76s %-10s calls %5s avg %8s vgpr %s"%(k,r["blocks"],r["calls"],r["avg_us"],r["vgpr"]))
PY
