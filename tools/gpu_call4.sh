#!/bin/bash
set -u
REPO=$PWD
OUT=$REPO/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
T0=$(date +%s)
echo "== GPU suite"
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider > "$OUT/c4_tests.log" 2>&1
tail -12 "$OUT/c4_tests.log" | cut -c1-220
echo "== bench ($(( $(date +%s) - T0 )) s)"
B="timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline"
$B --workload cfg2 > "$OUT/c4_bench_cfg2_default.json" 2> "$OUT/c4_bench_cfg2_default.err"
$B --workload cfg3_shard > "$OUT/c4_bench_cfg3s.json" 2> "$OUT/c4_bench_cfg3s.err"
$B --workload cfg5 > "$OUT/c4_bench_cfg5.json" 2> "$OUT/c4_bench_cfg5.err"
$B --workload cfg3 > "$OUT/c4_bench_cfg3.json" 2> "$OUT/c4_bench_cfg3.err"
for f in cfg2_default cfg3s cfg5 cfg3; do
  python - "$OUT/c4_bench_$f.json" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('c4_bench_')[1], round(d["value"]/1e6,3),"M/s ms",round(d["ms_per_step"],2),"nolog",round(d["ms_per_step_no_readback"],2),"grp_us",round(d["roofline"]["avg_launch_us"],1),"frac",round(d["roofline"]["frac"],3),{k:round(v,3) for k,v in d["phases_device_ms"].items()})
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
echo "== kernel traces ($(( $(date +%s) - T0 )) s)"
for WL in cfg2; do
  dir=/tmp/prof_$WL; rm -rf $dir
  (cd /tmp && timeout -s KILL 200 rocprofv3 --kernel-trace -d $dir -- python $REPO/bench.py --workload $WL --steps 3 --warmup 1 --no-cpu-baseline --set graph_update=False > "$OUT/c4_trace_$WL.log" 2>&1)
  DB=$(find $dir -name '*.db' | head -1)
  [ -n "$DB" ] && python tools/rocpd_stats.py "$DB" > "$OUT/c4_kernel_stats_$WL.csv"
done
grep -E "rollout_p|head_act|copyBuffer|64, 64, true, true" "$OUT/c4_kernel_stats_cfg2.csv" | cut -c1-60,150-260
echo "== done ($(( $(date +%s) - T0 )) s)"
