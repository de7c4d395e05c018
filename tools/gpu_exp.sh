#!/bin/bash
# A/B of run-time switches inside ONE gpurun call (box-to-box variance is +-3-5 %: only same-call comparisons count).
#   bash tools/gpu_exp.sh "<workload> <tag> [VAR=value ...]" ...        each argument = one bench run; all of them are
#                                                                        repeated ROUNDS (default 2) times, interleaved
#   EXTRA="--mlp-precision bf16x3" ...                                   extra bench.py arguments for every run
#   PYTEST="tests/test_gpu_kernels.py -k minibatch"                      optional parity run in front
# example:  bash tools/gpu_exp.sh "cfg2 two_launches CATPPO_FUSED_HEAD=0" "cfg2 fused" "cfg5 two_launches CATPPO_FUSED_HEAD=0" "cfg5 fused"
set -u
[ -n "${PYTEST:-}" ] && timeout 900 python -m pytest $PYTEST -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3
SPECS=("$@")
for r in $(seq 1 ${ROUNDS:-2}); do
  for spec in "${SPECS[@]}"; do
    set -- $spec; wl=$1; tag=$2; shift 2
    env "$@" X=1 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --workload $wl ${EXTRA:-} 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%-11s %-28s' % ('$wl','$tag'), round(d['value']/1e6,3),'M/s grp_us',round(d['roofline']['avg_launch_us'],1),'update_ms',round(d['phases_device_ms']['update_ms'],3),'rollout_ms',round(d['phases_device_ms']['rollout_ms'],3))"
  done
done
