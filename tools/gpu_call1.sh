#!/bin/bash
# first GPU call of round 2: new tests, A/B bench lines, kernel traces.  Everything lands in gpurun_out/.
set -u
OUT=$PWD/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
T0=$(date +%s)
echo "== r2 feature + parity tests" 
timeout 900 python -m pytest tests/test_gpu_r2_features.py tests/test_gpu_parity_sizes.py -m gpu -q --timeout 600 -p no:cacheprovider > "$OUT/c1_tests_new.log" 2>&1
tail -25 "$OUT/c1_tests_new.log"
echo "== old GPU suite ($(( $(date +%s) - T0 )) s)"
timeout 900 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider --deselect tests/test_gpu_r2_features.py --deselect tests/test_gpu_parity_sizes.py > "$OUT/c1_tests_old.log" 2>&1
tail -15 "$OUT/c1_tests_old.log"
echo "== bench A/B ($(( $(date +%s) - T0 )) s)"
B="timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline"
$B --workload cfg2 > "$OUT/c1_bench_cfg2_default.json" 2> "$OUT/c1_bench_cfg2_default.err"
$B --workload cfg2 --set graph_update=True > "$OUT/c1_bench_cfg2_graph.json" 2> "$OUT/c1_bench_cfg2_graph.err"
$B --workload cfg2 --set fused_rollout=False --set rng=torch > "$OUT/c1_bench_cfg2_r1path.json" 2> "$OUT/c1_bench_cfg2_r1path.err"
$B --workload cfg3_shard --set graph_update=False > "$OUT/c1_bench_cfg3s_eager.json" 2> "$OUT/c1_bench_cfg3s_eager.err"
$B --workload cfg3_shard --set graph_update=True > "$OUT/c1_bench_cfg3s_graph.json" 2> "$OUT/c1_bench_cfg3s_graph.err"
for f in cfg2_default cfg2_graph cfg2_r1path cfg3s_eager cfg3s_graph; do
  python - "$OUT/c1_bench_$f.json" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('c1_bench_')[1], round(d["value"]/1e6,3),"M/s ms",round(d["ms_per_step"],2),"nolog",round(d["ms_per_step_no_readback"],2),"grp_us",round(d["roofline"]["avg_launch_us"],1),"frac",round(d["roofline"]["frac"],3),d["phases_device_ms"], "gae", round(d["gae"]["config_size"]["us"],1), round(d["gae"]["config_size_scan"]["us"],1))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
echo "== kernel traces ($(( $(date +%s) - T0 )) s)"
for WL in cfg2 cfg3_shard; do
  dir=/tmp/prof_$WL; rm -rf $dir
  (cd /tmp && timeout -s KILL 200 rocprofv3 --kernel-trace -d $dir -- python $PWD/bench.py --workload $WL --steps 3 --warmup 1 --no-cpu-baseline --set graph_update=False > "$OUT/c1_trace_$WL.log" 2>&1)
  DB=$(find $dir -name '*.db' | head -1)
  [ -n "$DB" ] && python tools/rocpd_stats.py "$DB" > "$OUT/c1_kernel_stats_$WL.csv"
done
head -30 "$OUT/c1_kernel_stats_cfg3_shard.csv"
echo "== done ($(( $(date +%s) - T0 )) s)"
