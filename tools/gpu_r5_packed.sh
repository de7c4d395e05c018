#!/bin/bash
# round 5: bf16 planes split at staging time (GEMM_PACKED) - parity tests, then interleaved A/B against the per-use
# conversion of rounds 1-4 (tools/bin/libcatppo_unpacked.so = the same tree built with -DGEMM_PACKED=0), then traces
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_kernels.py -m gpu -q -x -p no:cacheprovider -k "bf16 or fused_head_launch" 2>&1 | tail -5
TREE=$PWD/constraints-as-terminations_amd/lib/libcatppo.so
OLD=$PWD/tools/bin/libcatppo_unpacked.so
ROUNDS=${ROUNDS:-2} EXTRA="--mlp-precision bf16x3" OUT=gpurun_out/r5_ab_packed_bf16x3.jsonl bash tools/gpu_exp.sh "cfg2 per_use CATPPO_LIB=$OLD" "cfg2 packed CATPPO_LIB=$TREE" > gpurun_out/r5_ab_packed_bf16x3.txt 2>&1
tail -3 gpurun_out/r5_ab_packed_bf16x3.txt
ROUNDS=${ROUNDS:-2} OUT=gpurun_out/r5_ab_packed_cfg5.jsonl bash tools/gpu_exp.sh "cfg5 per_use CATPPO_LIB=$OLD" "cfg5 packed CATPPO_LIB=$TREE" > gpurun_out/r5_ab_packed_cfg5.txt 2>&1
tail -3 gpurun_out/r5_ab_packed_cfg5.txt
BENCH_ARGS="--mlp-precision bf16x3" bash tools/gpu_trace_one.sh cfg2 r5pk_bf16x3 > /dev/null 2>&1
mv gpurun_out/r5pk_bf16x3_bench_cfg2_kernel_stats.csv gpurun_out/r5pk_bench_cfg2_bf16x3_kernel_stats.csv
bash tools/gpu_trace_one.sh cfg5 r5pk > /dev/null 2>&1
python - <<'PY'
import csv
for f in ("gpurun_out/r5pk_bench_cfg2_bf16x3_kernel_stats.csv","gpurun_out/r5pk_bench_cfg5_kernel_stats.csv"):
    print(f)
    for r in list(csv.DictReader(open(f)))[:9]:
        k=r["kernel"].replace("(anonymous namespace)::","").replace("void ","").split("(")[0][:74]
        print("  %-76s %-10s calls %5s avg %8s vgpr %s"%(k,r["blocks"],r["calls"],r["avg_us"],r["vgpr"]))
PY
