#!/bin/bash
# round 5: the tests the first call did not reach, the phase timeline of rollout_pre / rollout_post with the deferred
# tail, and where the kernel arguments live (HIP_FORCE_DEV_KERNARG) - the rollout is four ~4-14 us launches per env step
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
T0=$(date +%s)
timeout 900 python -m pytest tests/test_gpu_r5.py tests/test_gpu_rl_games.py tests/test_gpu_two_rank_trainer.py -m gpu -q --timeout 600 -p no:cacheprovider -k "not shape_generic and not norm_based and not adv_moments" > gpurun_out/rollout2_tests.log 2>&1
tail -4 gpurun_out/rollout2_tests.log | cut -c1-250
echo "== timeline ($(( $(date +%s) - T0 )) s)"
CATPPO_LIB=$PWD/tools/bin/libcatppo_tl.so timeout 200 python tools/rollout_timeline.py cfg2 2>&1 | grep -v "amdgpu.ids\|^\[INFO\]\|^Index\|^[0-9] |\|Active Constraint\|^$" > gpurun_out/r5_rollout_timeline.txt
cat gpurun_out/r5_rollout_timeline.txt
echo "== kernarg placement ($(( $(date +%s) - T0 )) s)"
ROUNDS=2 OUT=gpurun_out/r5_ab_kernarg.jsonl bash tools/gpu_exp.sh "cfg2 default X=1" "cfg2 dev_kernarg HIP_FORCE_DEV_KERNARG=1" "cfg2 host_kernarg HIP_FORCE_DEV_KERNARG=0" > gpurun_out/r5_ab_kernarg.txt 2>&1
tail -4 gpurun_out/r5_ab_kernarg.txt
echo "== done ($(( $(date +%s) - T0 )) s)"
