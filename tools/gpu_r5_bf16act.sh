#!/bin/bash
# round 5: H / dZ kept as bf16 in memory in the bf16-operand mode (BASELINE config 5): parity tests, interleaved A/B, trace
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_parity_sizes.py -m gpu -q -x -p no:cacheprovider -k "bf16 or cfg5" 2>&1 | tail -4
OLD=$PWD/tools/bin/libcatppo_unpacked.so
ROUNDS=2 OUT=gpurun_out/r5_ab_bf16_act.jsonl bash tools/gpu_exp.sh "cfg5 fp32_storage CATPPO_BF16_ACT=0" "cfg5 bf16_storage CATPPO_BF16_ACT=1" "cfg2 fp32_mode_before CATPPO_LIB=$OLD" "cfg2 fp32_mode_now X=1" > gpurun_out/r5_ab_bf16_act.txt 2>&1
tail -5 gpurun_out/r5_ab_bf16_act.txt
bash tools/gpu_trace_one.sh cfg5 r5act > /dev/null 2>&1
python - <<'PY'
import csv
for r in list(csv.DictReader(open("gpurun_out/r5act_bench_cfg5_kernel_stats.csv")))[:8]:
    k=r["kernel"].replace("(anonymous namespace)::","").replace("void ","").split("(")[0][:74]
    print("  %-76s %-10s calls %5s avg %8s vgpr %s"%(k,r["blocks"],r["calls"],r["avg_us"],r["vgpr"]))
PY
