#!/bin/bash
# one-call experiment runner: each line = tag + env settings
set -u
run() { local wl=$1; local tag=$2; shift 2; env "$@" timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --workload $wl ${EXTRA:-} 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%-11s %-28s' % ('$wl','$tag'), round(d['value']/1e6,3),'M/s grp_us',round(d['roofline']['avg_launch_us'],1),'update_ms',round(d['phases_device_ms']['update_ms'],3),'rollout_ms',round(d['phases_device_ms']['rollout_ms'],3))"; }
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_bf16.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3
for r in 1 2; do
for wl in cfg2 reference; do
run $wl base CATPPO_LIB=$PWD/tools/bin/libcatppo_base.so
run $wl new X=1
done
done
