#!/bin/bash
# round 5: first-layer weight gradient straight from global memory (dw0_direct_body) - bit-identity tests, A/B, trace
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_r4.py tests/test_gpu_kernels.py -m gpu -q -x -p no:cacheprovider -k "first_layer_weight or minibatch or one_call" 2>&1 | tail -3
TREE=$PWD/constraints-as-terminations_amd/lib/libcatppo.so
ROUNDS=3 OUT=gpurun_out/r5_ab_dw0_direct.jsonl bash tools/gpu_exp.sh "cfg2 gemm_body CATPPO_DW0_DIRECT=0" "cfg2 direct_d3 X=1" "cfg2 direct_d2 CATPPO_LIB=$PWD/tools/bin/libcatppo_dwd2.so" "cfg2 direct_d4 CATPPO_LIB=$PWD/tools/bin/libcatppo_dwd4.so" "reference gemm_body CATPPO_DW0_DIRECT=0" "reference direct_d3 X=1" > gpurun_out/r5_ab_dw0_direct.txt 2>&1
tail -7 gpurun_out/r5_ab_dw0_direct.txt
bash tools/gpu_trace_one.sh cfg2 r5dw > /dev/null 2>&1
python - <<'PY'
import csv
for r in list(csv.DictReader(open("gpurun_out/r5dw_bench_cfg2_kernel_stats.csv")))[:12]:
    k=r["kernel"].replace("(anonymous namespace)::","").replace("void ","").split("(")[0][:74]
    print("  %-76s %-10s calls %5s avg %8s vgpr %s"%(k,r["blocks"],r["calls"],r["avg_us"],r["vgpr"]))
PY
