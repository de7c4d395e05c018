#!/bin/bash
set -u
REPO=$PWD
OUT=$REPO/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
T0=$(date +%s)
echo "== GPU suite"
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider > "$OUT/c7_tests.log" 2>&1
tail -6 "$OUT/c7_tests.log" | cut -c1-220
echo "== strong-scaling shares ($(( $(date +%s) - T0 )) s)"
: > "$OUT/r2_bench_cfg3_shares.jsonl"
for W in 1 2 4 8; do
  timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --workload cfg3 --shard-of $W 2>/dev/null | tail -1 >> "$OUT/r2_bench_cfg3_shares.jsonl"
done
python - "$OUT/r2_bench_cfg3_shares.jsonl" <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if not l.strip(): continue
    d=json.loads(l); c=d["config"]
    print("share of", c["simulated_shard_of_world"], "envs", c["envs_per_gpu"], "mb", c["minibatch_per_gpu"], round(d["value"]/1e6,3),"M/s/rank ms",round(d["ms_per_step"],2),"grp_us",round(d["roofline"]["avg_launch_us"],1),"graph" if c["graph_update"] else "")
PY
echo "== microbench ($(( $(date +%s) - T0 )) s)"
timeout 600 python tools/microbench.py --reps 30 2>/dev/null | grep '^{' > "$OUT/r2_microbench.jsonl"
cut -c1-200 "$OUT/r2_microbench.jsonl"
echo "== cfg2 ($(( $(date +%s) - T0 )) s)"
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,3), d['ms_per_step'], d['phases_device_ms'], d['roofline']['avg_launch_us'])"
echo "== done ($(( $(date +%s) - T0 )) s)"
