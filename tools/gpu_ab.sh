#!/bin/bash
# A/B of library builds inside one gpurun call: CATPPO_LIB selects the .so.
#   tools/gpu_ab.sh [lib-name ...]     names under tools/bin/libcatppo_<name>.so; "tree" = the in-tree build
# WLS="cfg2 reference" ROUNDS=2 select workloads / repetitions.
set -u
LIBS=${@:-base tree}
for round in $(seq 1 ${ROUNDS:-2}); do
  for WL in ${WLS:-cfg2 reference}; do
    for name in $LIBS; do
      if [ "$name" = tree ]; then lib=$PWD/constraints-as-terminations_amd/lib/libcatppo.so; else lib=$PWD/tools/bin/libcatppo_$name.so; fi
      CATPPO_LIB=$lib timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --workload $WL 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%-10s %-12s' % ('$WL', '$name'), round(d['value']/1e6,3),'M/s grp_us',round(d['roofline']['avg_launch_us'],1),'update_ms',round(d['phases_device_ms']['update_ms'],3),'rollout_ms',round(d['phases_device_ms']['rollout_ms'],3))"
    done
  done
done
