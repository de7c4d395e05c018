#!/bin/bash
set -u
REPO=$PWD
OUT=$REPO/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
B="timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --workload cfg2"
run() {  # tag, env...
  local tag=$1; shift
  env "$@" $B 2>/dev/null | tail -1 > "$OUT/c6_$tag.json"
  python - "$OUT/c6_$tag.json" "$tag" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(f'{sys.argv[2]:28s}', round(d["value"]/1e6,3),"M/s grp_us",round(d["roofline"]["avg_launch_us"],1),"update_ms",round(d["phases_device_ms"]["update_ms"],3))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
run default X=1
run t64x128 CATPPO_FWD_TILE=64x128
run t64x128_pad40k CATPPO_FWD_TILE=64x128 CATPPO_FWD_LDS_PAD=40000
run t64x128_pad20k CATPPO_FWD_TILE=64x128 CATPPO_FWD_LDS_PAD=20000
run t64x64 CATPPO_FWD_TILE=64x64
run t64x64_pad20k CATPPO_FWD_TILE=64x64 CATPPO_FWD_LDS_PAD=20000
run t64x64_pad60k CATPPO_FWD_TILE=64x64 CATPPO_FWD_LDS_PAD=60000
run t128x64 CATPPO_FWD_TILE=128x64
run t128x64_pad40k CATPPO_FWD_TILE=128x64 CATPPO_FWD_LDS_PAD=40000
run t128x128_pad45k CATPPO_FWD_LDS_PAD=45000
run default2 X=1
