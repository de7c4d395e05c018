#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_r4.py tests/test_gpu_parity_sizes.py -m gpu -q -x -p no:cacheprovider -k "bf16 or cfg5 or first_layer" 2>&1 | tail -3
ROUNDS=3 OUT=gpurun_out/r5_ab_dwfold_bf16.jsonl bash tools/gpu_exp.sh "cfg5 own_launch CATPPO_DW0_FOLD=0" "cfg5 with_fold X=1" > gpurun_out/r5_ab_dwfold_bf16.txt 2>&1
tail -3 gpurun_out/r5_ab_dwfold_bf16.txt
ROUNDS=3 EXTRA="--mlp-precision bf16x3" OUT=gpurun_out/r5_ab_dwfold_bf16x3.jsonl bash tools/gpu_exp.sh "cfg2 own_launch CATPPO_DW0_FOLD=0" "cfg2 with_fold X=1" > gpurun_out/r5_ab_dwfold_bf16x3.txt 2>&1
tail -3 gpurun_out/r5_ab_dwfold_bf16x3.txt
