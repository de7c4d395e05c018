#!/bin/bash
# round 3, GPU call A: new multi-rank / parity tests, the rest of the suite, XCD-mapping A/B, traffic passes
set -u
OUT=$PWD/gpurun_out; mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_two_rank_trainer.py -x -q -s > "$OUT/a_two_rank.log" 2>&1; echo "two_rank rc=$?"
tail -5 "$OUT/a_two_rank.log"
timeout 1500 python -m pytest tests -m gpu -q --deselect tests/test_gpu_two_rank_trainer.py --durations=8 > "$OUT/a_suite.log" 2>&1; echo "suite rc=$?"
tail -15 "$OUT/a_suite.log"
for L in 1 0; do
  CATPPO_XCD_LEGACY=$L python bench.py --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/a_bench_cfg2_xcdlegacy$L.json" 2> "$OUT/a_bench_cfg2_xcdlegacy$L.err"
  python - <<PY
import json
d=json.load(open("$OUT/a_bench_cfg2_xcdlegacy$L.json"))
print("legacy=$L", round(d["value"]/1e6,3), "M/s", round(d["ms_per_step"],3), "ms group_us", round(d["roofline"]["avg_launch_us"],1), d["phases_device_ms"])
PY
done
CATPPO_XCD_LEGACY=1 python bench.py --workload reference --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ref legacy=1', d['value']/1e6, d['roofline']['avg_launch_us'])"
python bench.py --workload reference --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ref legacy=0', d['value']/1e6, d['roofline']['avg_launch_us'])"
SQ_PASSES=0 bash tools/profile_bench.sh cfg2 r3 > "$OUT/a_profile.log" 2>&1
cat "$OUT/r3_pmc_traffic_cfg2.json" | head -40
