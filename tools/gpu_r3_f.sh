#!/bin/bash
set -u
OUT=$PWD/gpurun_out; mkdir -p "$OUT"
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_r2_features.py tests/test_gpu_end_to_end.py -x -q > "$OUT/f_tests.log" 2>&1; echo "tests rc=$?"; tail -6 "$OUT/f_tests.log"
for F in 0 1; do
  for WL in cfg2 cfg3_shard; do
  CATPPO_FUSED_FWD=$F python bench.py --workload $WL --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('fused_fwd=$F $WL', round(d['value']/1e6,3),'M/s ms',round(d['ms_per_step'],3),'phases',{k:round(v,3) for k,v in d['phases_device_ms'].items()})"
  done
done
