#!/bin/bash
set -u
OUT=$PWD/gpurun_out; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_r6.py tests/test_gpu_kernels.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -5
L=$PWD/tools/bin
EXTRA="--shard-of 8" ROUNDS=2 OUT=$OUT/r6_ab_stage_partial_cfg3.jsonl bash tools/gpu_exp.sh "cfg3 staged" "cfg3 dword_stores CATPPO_LIB=$L/libcatppo_nostage.so" 2>&1 | tee $OUT/r6_ab_stage_partial.txt
ROUNDS=2 OUT=$OUT/r6_ab_stage_partial_cfg2.jsonl bash tools/gpu_exp.sh "cfg2 staged" "cfg2 dword_stores CATPPO_LIB=$L/libcatppo_nostage.so" "reference staged" "reference dword_stores CATPPO_LIB=$L/libcatppo_nostage.so" 2>&1 | tee -a $OUT/r6_ab_stage_partial.txt
