#!/bin/bash
# A/B of two library builds inside one call: CATPPO_LIB selects the .so
set -u
OUT=$PWD/gpurun_out; mkdir -p "$OUT"
BASE=$PWD/constraints-as-terminations_amd/lib/libcatppo_base.so
run() { local tag=$1; shift; env "$@" timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --workload ${WL:-cfg2} 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$tag', round(d['value']/1e6,3),'M/s grp_us',round(d['roofline']['avg_launch_us'],1),'update_ms',round(d['phases_device_ms']['update_ms'],3),'rollout_ms',round(d['phases_device_ms']['rollout_ms'],3))"; }
for WL in cfg2 cfg5 reference; do
  export WL
  run "$WL base" CATPPO_LIB=$BASE
  run "$WL new " X=1
  run "$WL base" CATPPO_LIB=$BASE
  run "$WL new " X=1
done
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "minibatch" -p no:cacheprovider 2>&1 | tail -2
