#!/bin/bash
set -u
run() { local wl=$1; local tag=$2; shift 2; env "$@" timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --workload $wl ${EXTRA:-} 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%-11s %-28s' % ('$wl','$tag'), round(d['value']/1e6,3),'M/s grp_us',round(d['roofline']['avg_launch_us'],1),'update_ms',round(d['phases_device_ms']['update_ms'],3),'rollout_ms',round(d['phases_device_ms']['rollout_ms'],3))"; }
for r in 1 2; do
EXTRA="--shard-of 4" run cfg3 W4_default X=1
EXTRA="--shard-of 4" run cfg3 W4_fused CATPPO_FUSED_HEAD_MIN_WG=64
EXTRA="--shard-of 8" run cfg3 W8_default X=1
EXTRA="--shard-of 8" run cfg3 W8_fused CATPPO_FUSED_HEAD_MIN_WG=64
done
