#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_kernels.py tests/test_gpu_r2_features.py -m gpu -q -x -p no:cacheprovider -k "bf16 or fused_head_launch" 2>&1 | tail -3
TREE=$PWD/constraints-as-terminations_amd/lib/libcatppo.so
OLD=$PWD/tools/bin/libcatppo_unpacked.so
ROUNDS=3 EXTRA="--mlp-precision bf16x3" OUT=gpurun_out/r5_ab_packed_bf16x3.jsonl bash tools/gpu_exp.sh "cfg2 per_use CATPPO_LIB=$OLD" "cfg2 packed CATPPO_LIB=$TREE" "reference per_use CATPPO_LIB=$OLD" "reference packed CATPPO_LIB=$TREE" > gpurun_out/r5_ab_packed_bf16x3.txt 2>&1
tail -5 gpurun_out/r5_ab_packed_bf16x3.txt
