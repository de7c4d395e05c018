#!/bin/bash
# round 6: step16 parity tests, phase timeline, prefetch depth variants (interleaved A/B at one rank's share of cfg3)
set -u
OUT=$PWD/gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_r6.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -5
for v in ${TL_VARIANTS:-s16tl}; do
  for net in ref cfg2; do
    echo "==== $v $net"; CATPPO_LIB=$PWD/tools/bin/libcatppo_$v.so python tools/step16_timeline.py 2048 $net 2>&1 | grep -v amdgpu.ids
  done
done | tee $OUT/r6_step16_timeline.txt
L=$PWD/tools/bin
SPECS=("cfg3 default")
for v in ${VARIANTS:-}; do SPECS+=("cfg3 $v CATPPO_LIB=$L/libcatppo_$v.so"); done
EXTRA="--shard-of 8" ROUNDS=2 OUT=$OUT/r6_ab_step16_depth.jsonl bash tools/gpu_exp.sh "${SPECS[@]}" 2>&1 | tee $OUT/r6_ab_step16_depth.txt
