#!/bin/bash
set -u
REPO=$PWD
OUT=$REPO/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
T0=$(date +%s)
echo "== GPU suite"
timeout 1200 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider -x > "$OUT/c2_tests.log" 2>&1
tail -30 "$OUT/c2_tests.log"
echo "== bench A/B ($(( $(date +%s) - T0 )) s)"
B="timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline"
$B --workload cfg2 > "$OUT/c2_bench_cfg2_default.json" 2> "$OUT/c2_bench_cfg2_default.err"
$B --workload cfg2 --mlp-precision bf16x3 > "$OUT/c2_bench_cfg2_bf16x3.json" 2> "$OUT/c2_bench_cfg2_bf16x3.err"
$B --workload cfg2 --mlp-precision bf16 > "$OUT/c2_bench_cfg2_bf16.json" 2> "$OUT/c2_bench_cfg2_bf16.err"
$B --workload reference > "$OUT/c2_bench_reference.json" 2> "$OUT/c2_bench_reference.err"
$B --workload cfg3_shard --mlp-precision bf16x3 > "$OUT/c2_bench_cfg3s_bf16x3.json" 2> "$OUT/c2_bench_cfg3s_bf16x3.err"
for f in cfg2_default cfg2_bf16x3 cfg2_bf16 reference cfg3s_bf16x3; do
  python - "$OUT/c2_bench_$f.json" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('c2_bench_')[1], round(d["value"]/1e6,3),"M/s ms",round(d["ms_per_step"],2),"nolog",round(d["ms_per_step_no_readback"],2),"grp_us",round(d["roofline"]["avg_launch_us"],1),"frac",round(d["roofline"]["frac"],3),d["phases_device_ms"])
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
echo "== host profile ($(( $(date +%s) - T0 )) s)"
timeout 200 python tools/host_profile.py > "$OUT/c2_host_profile.txt" 2>&1
head -45 "$OUT/c2_host_profile.txt"
timeout 100 python tools/phase_times.py --workload cfg2 > "$OUT/c2_phase_times_cfg2.json" 2>&1; cat "$OUT/c2_phase_times_cfg2.json"
echo "== kernel traces ($(( $(date +%s) - T0 )) s)"
for WL in cfg2 cfg3_shard; do
  dir=/tmp/prof_$WL; rm -rf $dir
  (cd /tmp && timeout -s KILL 200 rocprofv3 --kernel-trace -d $dir -- python $REPO/bench.py --workload $WL --steps 3 --warmup 1 --no-cpu-baseline --set graph_update=False > "$OUT/c2_trace_$WL.log" 2>&1)
  DB=$(find $dir -name '*.db' | head -1)
  [ -n "$DB" ] && python tools/rocpd_stats.py "$DB" > "$OUT/c2_kernel_stats_$WL.csv"
done
dir=/tmp/prof_x3; rm -rf $dir
(cd /tmp && timeout -s KILL 200 rocprofv3 --kernel-trace -d $dir -- python $REPO/bench.py --workload cfg2 --mlp-precision bf16x3 --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/c2_trace_cfg2_bf16x3.log" 2>&1)
DB=$(find $dir -name '*.db' | head -1)
[ -n "$DB" ] && python tools/rocpd_stats.py "$DB" > "$OUT/c2_kernel_stats_cfg2_bf16x3.csv"
cut -c1-150 "$OUT/c2_kernel_stats_cfg2.csv" | head -24
echo "== done ($(( $(date +%s) - T0 )) s)"
