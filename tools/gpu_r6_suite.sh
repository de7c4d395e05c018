#!/bin/bash
# the whole GPU suite + smoke + A/B of the 16-row rollout forward
set -u
OUT=$PWD/gpurun_out; mkdir -p $OUT
timeout 2400 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | tail -8 | tee $OUT/r6_gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
EXTRA="--shard-of 8" ROUNDS=2 OUT=$OUT/r6_ab_step16_fwd.jsonl bash tools/gpu_exp.sh "cfg3 layerwise_rollout CATPPO_STEP16_FWD=0" "cfg3 step16_fwd" "cfg3 step16_fwd_4096 CATPPO_STEP16_FWD_MAX_ROWS=4096" 2>&1 | tee $OUT/r6_ab_step16_fwd.txt
ROUNDS=2 OUT=$OUT/r6_ab_step16_fwd_cfg2.jsonl bash tools/gpu_exp.sh "cfg2 rows32" "cfg2 step16_fwd_4096 CATPPO_STEP16_FWD_MAX_ROWS=4096" "reference rows32" "reference step16_fwd_4096 CATPPO_STEP16_FWD_MAX_ROWS=4096" 2>&1 | tee -a $OUT/r6_ab_step16_fwd.txt
