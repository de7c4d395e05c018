#!/bin/bash
# round 5: rollout_pre with the state copy's stores behind the terms, rollout_post with the operands of its state
# derivation requested up front - parity tests of everything that steps an env, A/B against the previous build, timeline
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
T0=$(date +%s)
timeout 1200 python -m pytest tests/test_gpu_end_to_end.py tests/test_gpu_r2_features.py tests/test_gpu_r4.py tests/test_gpu_r5.py tests/test_gpu_two_rank_trainer.py tests/test_gpu_parity_sizes.py -m gpu -q --timeout 600 -p no:cacheprovider -x -k "not shape_generic and not adv_moments and not gradient" > gpurun_out/rollout3_tests.log 2>&1
tail -4 gpurun_out/rollout3_tests.log | cut -c1-250
echo "== A/B ($(( $(date +%s) - T0 )) s)"
OLD=$PWD/tools/bin/libcatppo_r5d.so
ROUNDS=3 OUT=gpurun_out/r5_ab_rollout_loads.jsonl bash tools/gpu_exp.sh "cfg2 previous CATPPO_LIB=$OLD" "cfg2 loads_up_front X=1" "reference previous CATPPO_LIB=$OLD" "reference loads_up_front X=1" > gpurun_out/r5_ab_rollout_loads.txt 2>&1
tail -5 gpurun_out/r5_ab_rollout_loads.txt
echo "== timeline ($(( $(date +%s) - T0 )) s)"
CATPPO_LIB=$PWD/tools/bin/libcatppo_tl.so timeout 200 python tools/rollout_timeline.py cfg2 2>&1 | grep -v "amdgpu.ids\|^\[INFO\]\|^Index\|^[0-9] |\|Active Constraint\|^$" > gpurun_out/r5_rollout_timeline2.txt
cat gpurun_out/r5_rollout_timeline2.txt
bash tools/gpu_trace_one.sh cfg2 r5loads > /dev/null 2>&1
grep -i "rollout" gpurun_out/r5loads_bench_cfg2_kernel_stats.csv | cut -d, -f1-8 | sed 's/(anonymous namespace):://' | cut -c1-120
echo "== done ($(( $(date +%s) - T0 )) s)"
